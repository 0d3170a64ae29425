"""(f)4 with coupled joint dynamics: MI_MODEL_ARM27C - the 7-joint arm of MI_MODEL_ARM27 with the manipulator equation
M(q) qdd = tau - b qd - J^T F - sum m_p J_p^T (a_p + g e_z) over three point masses and rotor inertias (csrc/models.hpp:
Arm27C; oracle/models_np.py: arm27c_step) pushing the free ball: the state stack, horizon, cost, target, delta and beta of
kinova_gen3.py:52-70,254-284, a dense 7x7 mass matrix solved per step and fx without ARM27's block structure.  On the mid-size
workgroup-per-problem kernels (ilqr_large.hpp: mid_backward) against the six arm27c_* fixtures recorded from the UNMODIFIED
reference (oracle/gen_golden.py) and, at batch scale, against the C oracle (pinned to the same fixtures by
tests/test_c_oracle.py).  All through the C ABI."""
import numpy as np
import pytest

from common import assert_flip_budget, load_golden, rel_err
from test_gpu_parity import make_solver
from test_gpu_arm27 import _own_mpc_sensitivity

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("jac", ["ad", "fd"])
def test_arm27c_stage_level_vs_reference_golden(jac):
    """The third iteration's stages against the snapshot of the unmodified reference (arm27c_stage), tolerances of SURVEY 8(c)
    with identical inputs: rollout 1e-10; Jacobians 1e-10 with duals (27 + 7 derivative lanes through the velocity-product
    recursion and the LDL^T solve), 2e-6 with central differences (contact curvature k / sigma^2 = 6e7); gains against the
    reference's fp64 pass and the extended-precision pass of the same inputs."""
    from common import backward_errors, make_oracle
    g, prob = load_golden("arm27c_stage")
    s = make_solver(prob, jac=jac)
    s.SetInitialState(g["x0"][None])
    s.SetInitialGuess(g["pre_u_bar"])
    s.set_state(x_bar=g["pre_x_bar"][None], K=g["pre_K"][None], kappa=g["pre_kappa"][None], dV_coeff=g["pre_dV"][None])
    x, u, L, ex = s.stage_rollout(1.0)
    assert rel_err(x[0], g["roll_x"]) < 1e-10 and rel_err(u[0], g["roll_u"]) < 1e-10
    assert abs(L[0] - g["roll_L"]) < 1e-10 * abs(g["roll_L"])
    s.set_state(x_bar=x, u_bar=u)
    s.stage_linearize()
    tolj = 1e-10 if jac == "ad" else 2e-6
    e_fx, e_fu = rel_err(s.fx[0], g["fx"]), rel_err(s.fu[0], g["fu"])
    print(f"arm27c Jacobians ({jac}): fx {e_fx:.2e} fu {e_fu:.2e}")
    assert e_fx < tolj and e_fu < tolj
    if jac == "fd":
        return
    s.set_state(x_bar=g["roll_x"][None], u_bar=g["roll_u"][None], fx=g["fx"][None], fu=g["fu"][None])
    s.stage_backward()
    o = make_oracle(prob)
    o.set_problem(g["x0"], prob["x_nom"], prob["Q"], prob["R"], prob["Qf"], g["roll_u"])
    o.x_bar, o.fx, o.fu = g["roll_x"], g["fx"], g["fu"]
    o.backward()
    e_dev, e_ref, cond = backward_errors((s.K[0], s.kappa[0], s.dV_coeff[0]), o)
    print(f"arm27c backward pass: device {e_dev:.2e}, NumPy fp64 {e_ref:.2e} from the extended-precision pass; max cond(Quu) {cond:.1e}")
    assert e_dev < max(1e-11, 20 * e_ref)
    assert rel_err(s.K[0], g["post_K"]) < 1e-9 and rel_err(s.kappa[0], g["post_kappa"]) < 1e-9 and rel_err(s.dV_coeff[0], g["post_dV"]) < 1e-9


@pytest.mark.parametrize("name", ["arm27c_solve_0", "arm27c_solve_1"])
def test_arm27c_solve_vs_reference_golden(name):
    """Whole solves recorded from the unmodified reference (exact Jacobians both sides): 18 iterations from the nominal start
    (arm27c_solve_0), 14 from a moved arm and ball (arm27c_solve_1).  Iterations, step sizes, trial counts exact; converged
    arrays to SURVEY 8(c)'s end-to-end figures or 5 x the reference algorithm's own one-ulp sensitivity, whichever is larger."""
    from common import make_oracle
    g, prob = load_golden(name)
    s = make_solver(prob, jac="ad", single=True, hist_cap=32)
    s.SetInitialState(g["x0"])
    s.SetInitialGuess(g["u_guess"])
    x, u, _, L = s.Solve()
    iters = int(s.iterations[0])
    assert iters == len(g["hist"])
    h = s.history[0][:iters]
    assert np.array_equal(h[:, 1:3], g["hist"][:, 1:3])
    assert rel_err(h[:, 0], g["hist"][:, 0]) < 1e-9 and abs(L - g["L"]) < 1e-9 * abs(g["L"])
    own = dict(K=0.0, x=0.0, u=0.0)
    base = None
    for d in (0.0, np.inf, -np.inf):
        o = make_oracle(prob)
        x0 = g["x0"].copy()
        if d:
            x0[0], x0[12] = np.nextafter(x0[0], d), np.nextafter(x0[12], d)
        o.set_problem(x0, prob["x_nom"], prob["Q"], prob["R"], prob["Qf"], g["u_guess"])
        xo, uo, _, _ = o.solve()
        if base is None:
            base = (o.K.copy(), xo.copy(), uo.copy())
            assert np.array_equal(xo, g["x_bar"])                     # (the NumPy oracle IS the reference on this fixture)
        else:
            own = dict(K=max(own["K"], rel_err(o.K, base[0])), x=max(own["x"], float(np.max(np.abs(xo - base[1])))), u=max(own["u"], float(np.max(np.abs(uo - base[2])))))
    eK, ex_, eu = rel_err(s.K, g["K"]), float(np.max(np.abs(x - g["x_bar"]))), float(np.max(np.abs(u - g["u_bar"])))
    print(f"{name}: device vs reference K {eK:.2e} x {ex_:.2e} u {eu:.2e}; the oracle one ulp away from itself K {own['K']:.2e} x {own['x']:.2e} u {own['u']:.2e}")
    assert ex_ < max(1e-8, 5 * own["x"]) and eu < max(1e-8, 5 * own["u"]) and eK < max(1e-7, 5 * own["K"])
    assert rel_err(s.fx, g["fx"]) < 1e-8


@pytest.mark.parametrize("device_loop", [False, True])
def test_arm27c_mpc_vs_reference_golden(device_loop):
    """Receding-horizon sequence (replan 5 steps, two re-solves) recorded from the reference with its persistent gains (SURVEY
    F10): host loop of Solve() calls, and the whole loop in one launch (mi_ilqr_mpc_run).  Iterations of every solve exact;
    cold solve's cost 1e-9; re-solves' costs to 5 x the reference algorithm's own one-ulp sensitivity."""
    from drake_ddp_amd.workloads import mpc_shift, arm27c_u_guess
    g, prob = load_golden("arm27c_mpc_0")
    s = make_solver(prob, jac="ad")
    N, replan, R = prob["N"], int(g["replan"]), len(g["Ls"]) - 1
    base, own, own_x, own_K = _own_mpc_sensitivity(prob, g["x0"], arm27c_u_guess(N), R, replan)
    assert np.array_equal(base, g["Ls"])
    s.SetInitialState(g["x0"][None])
    s.SetInitialGuess(arm27c_u_guess(N))
    x, u, _, L = s.Solve()
    assert s.iterations[0] == g["iters"][0] and abs(L[0] - g["Ls"][0]) < 1e-9 * abs(g["Ls"][0])
    if device_loop:
        s.MPCRun(R, replan)
        log = s.mpc_log[0]
        assert np.array_equal(log[:, -1].astype(int), g["iters"][1:])
        dev = np.abs(log[:, -2] - g["Ls"][1:]) / g["Ls"][1:]
    else:
        dev = []
        for r in range(1, R + 1):
            x0, ug = mpc_shift(x, u, replan)
            s.SetInitialState(x0); s.SetInitialGuess(ug)
            x, u, _, L = s.Solve()
            assert s.iterations[0] == g["iters"][r]
            dev.append(abs(L[0] - g["Ls"][r]) / g["Ls"][r])
        dev = np.array(dev)
    ex_, eK = rel_err(s.x_bar[0], g["xs"][-1]), rel_err(s.K[0], g["Ks"][-1])
    print(f"arm27c MPC re-solve costs: device vs reference {dev}, the reference one ulp away from itself {own[1:]}; "
          f"last x_bar {ex_:.2e} (own {own_x:.2e}), K {eK:.2e} (own {own_K:.2e})")
    assert np.all(dev < np.maximum(1e-8, 5 * own[1:]))
    assert ex_ < max(1e-7, 5 * own_x) and eK < max(1e-6, 5 * own_K)


@pytest.mark.parametrize("name", ["arm27c_kp_adaptivejerk", "arm27c_kp_iterativeerror"])
def test_arm27c_keypoint_methods_vs_reference_golden(name):
    """kinova_gen3.py:34-40's derivative interpolation on the coupled arm, recorded from the reference (adaptiveJerk minN 5,
    maxN 40, threshold 1e-4; iterativeError minN 5, threshold 1e-2).  Iterations, step sizes, trial counts, percentage of
    derivatives per iteration and the last key-point list exact; interpolated fx / fu 1e-7; cost 1e-8; converged arrays to
    SURVEY 8(c) or 10 x the C oracle's own one-ulp movement."""
    from common import golden_keypoint
    from oracle import c_oracle, models_np as M
    g, prob = load_golden(name)
    kp = golden_keypoint(g)
    s = make_solver(prob, keypoint=kp, jac="ad", single=True, hist_cap=32)
    s.SetInitialState(g["x0"])
    s.SetInitialGuess(g["u_guess"])
    x, u, _, L = s.Solve()
    iters = int(s.iterations[0])
    assert iters == len(g["hist"])
    h = s.history[0][:iters]
    assert np.array_equal(h[:, 1:3], g["hist"][:, 1:3]) and np.allclose(h[:, 3], g["hist"][:, 3], rtol=0, atol=1e-9)
    nk = int(s.keypoint_count[0])
    assert np.array_equal(s.keypoint_list[0][:nk], g["kp_last"])
    assert abs(L - g["L"]) <= 1e-8 * abs(g["L"]) and rel_err(h[:, 0], g["hist"][:, 0]) < 1e-8
    assert rel_err(s.fx, g["fx"]) < 1e-7 and rel_err(s.fu, g["fu"]) < 1e-7
    model = M.Model(prob["model_id"], prob["dt"], prob.get("params"))
    r0 = c_oracle.solve_batch(model, prob, g["x0"][None], g["u_guess"], keypoint=kp)
    own_x = own_K = 0.0
    for d in (np.inf, -np.inf):
        xq = g["x0"][None].copy()
        xq[:, 0], xq[:, 12] = np.nextafter(xq[:, 0], d), np.nextafter(xq[:, 12], d)
        rq = c_oracle.solve_batch(model, prob, xq, g["u_guess"], keypoint=kp)
        if rq["iters"][0] == r0["iters"][0]:
            own_x, own_K = max(own_x, float(np.max(np.abs(rq["x_bar"] - r0["x_bar"])))), max(own_K, rel_err(rq["K"], r0["K"]))
    e_x, e_K = float(np.max(np.abs(x - g["x_bar"]))), rel_err(s.K, g["K"])
    print(f"{name}: |x - x_golden| {e_x:.2e} (the oracle one ulp away from itself: {own_x:.2e}), K {e_K:.2e} ({own_K:.2e}); {nk} key points")
    assert e_x < max(1e-8, 10 * own_x) and e_K < max(1e-7, 10 * own_K)


@pytest.mark.parametrize("B", [1, 64])
def test_arm27c_batch_fd_vs_c_oracle(B):
    """B = 1 and B = 64 seeded problems (arm and ball moved), central differences on both sides, against the C oracle: status,
    iterations and line-search trials of every problem; costs and trajectories to the central-difference tolerances of 8(c)."""
    from drake_ddp_amd import workloads as W
    from oracle import c_oracle, models_np as M
    prob = W.arm27c_problem()
    x0, ug = W.arm27_batch_x0(64)[:B], W.arm27c_u_guess(prob["N"])
    s = make_solver(prob, B=B, jac="fd", hist_cap=64)
    s.SetInitialState(x0)
    s.SetInitialGuess(ug)
    x, u, _, L = s.Solve()
    model = M.Model(prob["model_id"], prob["dt"])
    r = c_oracle.solve_batch(model, prob, x0, ug, hist_cap=64)
    assert np.array_equal(s.status, r["status"]) and (s.status == 0).all()
    same = (s.iterations == r["iters"]) & (s.ls_trials == r["ls"])
    assert_flip_budget("arm27c_batch", same, (s.iterations[~same], r["iters"][~same]))
    rel = np.abs(L - r["cost"]) / np.abs(r["cost"])
    print(f"arm27c B={B}: iterations {int(s.iterations.sum())}, trials {int(s.ls_trials.sum())}, worst cost error {rel[same].max():.2e}, "
          f"worst |x - x_oracle| {np.max(np.abs(x[same] - r['x_bar'][same])):.2e}")
    assert np.max(rel[same]) < 5e-8 and np.max(np.abs(x[same] - r["x_bar"][same])) < 1e-6
    xq = x0.copy()
    xq[:, 0], xq[:, 12] = np.nextafter(xq[:, 0], np.inf), np.nextafter(xq[:, 12], -np.inf)
    rq = c_oracle.solve_batch(model, prob, xq, ug)
    keep = same & (rq["iters"] == r["iters"]) & (rq["ls"] == r["ls"])
    own_K, e_K = rel_err(rq["K"][keep], r["K"][keep]), rel_err(s.K[keep], r["K"][keep])
    print(f"arm27c B={B}: K device vs oracle {e_K:.2e}, oracle vs itself one ulp away {own_K:.2e} ({int(keep.sum())} problems)")
    assert e_K < max(1e-6, 10 * own_K)


def test_arm27c_mpc_run_vs_c_oracle():
    """The benchmarked coupled-arm config (C6b): B = 64, cold solve + MPCRun(20, 5) in one launch, every problem and re-solve
    against the C oracle's receding-horizon loop, with the yardsticks of test_arm27_mpc_run_vs_c_oracle (the C oracle against
    itself with x0 one ulp away)."""
    from drake_ddp_amd import workloads as W
    from oracle import c_oracle, models_np as M
    q = W.arm27c_problem()
    B, R, replan = 64, 20, 5
    x0, ug = W.arm27_batch_x0(B), W.arm27c_u_guess(q["N"])
    s = make_solver(q, B=B, jac="fd")
    s.SetInitialState(x0)
    s.SetInitialGuess(ug)
    s.Solve()
    first_it, first_L = s.iterations.copy(), s.cost.copy()
    st = s.MPCRun(R, replan)
    log = s.mpc_log
    model = M.Model(q["model_id"], q["dt"])
    r = c_oracle.mpc_batch(model, q, x0, ug, R, replan)
    same0 = first_it == r["first"][:, 1].astype(int)
    assert_flip_budget("arm27c_mpc_first", same0, (first_it[~same0], r["first"][:, 1][~same0]))
    assert np.max((np.abs(first_L - r["first"][:, 0]) / r["first"][:, 0])[same0]) < 5e-8
    assert (s.status == 0).all() and (r["status"] == 0).all() and st.n_converged == B
    own_flips, own_dev, own_dev_flipped, own_state = 0, np.zeros(R), 0.0, 0.0
    for d in (np.inf, -np.inf):
        xq = x0.copy()
        xq[:, 0], xq[:, 12] = np.nextafter(xq[:, 0], d), np.nextafter(xq[:, 12], d)
        rq = c_oracle.mpc_batch(model, q, xq, ug, R, replan)
        keep = (rq["log"][:, :, -1] == r["log"][:, :, -1]).all(axis=1)
        own_flips = max(own_flips, int((~keep).sum()))
        dq = np.abs(rq["log"][:, :, -2] - r["log"][:, :, -2]) / r["log"][:, :, -2]
        own_dev = np.maximum(own_dev, dq[keep].max(axis=0))
        if (~keep).any():
            own_dev_flipped = max(own_dev_flipped, float(dq[~keep].max()))
        own_state = max(own_state, float(np.max(np.abs(rq["log"][keep][:, :, :27] - r["log"][keep][:, :, :27]))))
    full = (log[:, :, -1] == r["log"][:, :, -1]).all(axis=1) & same0
    dev = (np.abs(log[:, :, -2] - r["log"][:, :, -2]) / r["log"][:, :, -2])
    print(f"arm27c MPC x{R}: device takes other iteration counts in {int((~full).sum())} of {B} problems (the oracle one ulp away from itself: {own_flips}); "
          f"worst re-solve cost deviation {dev[full].max():.2e} (the oracle's own: {own_dev.max():.2e})")
    assert int((~full).sum()) <= own_flips + 2
    assert np.all(dev[full].max(axis=0) < np.maximum(1e-7, 10 * own_dev))
    e_flip = float(dev[~full].max()) if (~full).any() else 0.0
    e_state = float(np.max(np.abs(log[full][:, :, :27] - r["log"][full][:, :, :27])))
    print(f"arm27c MPC: cost deviation of problems on another path {e_flip:.2e} (the oracle's own: {own_dev_flipped:.2e}); re-solve start states {e_state:.2e} (own {own_state:.2e})")
    assert e_flip <= max(1e-3, 10 * own_dev_flipped)
    assert e_state < max(1e-6, 10 * own_state)


def test_arm27c_through_the_reference_class_surface(tmp_path):
    """kinova_gen3.py:254-284's solver section on the drop-in class with the coupled arm's system object: constructor keywords,
    setters, Solve() tuple, SaveSolution's npz keys / shapes (ilqr.py:712-733)."""
    from drake_ddp_amd import workloads as W
    from drake_ddp_amd.ilqr import IterativeLinearQuadraticRegulator
    from drake_ddp_amd.models import ArmAndBallCoupled
    p = W.arm27c_problem()
    ilqr = IterativeLinearQuadraticRegulator(ArmAndBallCoupled(p["dt"]), p["N"], beta=0.5, delta=1e-3, gamma=0, derivs_keypoint_method=None, verbose=False)
    ilqr.SetInitialState(W.arm27_start())
    ilqr.SetTargetState(p["x_nom"])
    ilqr.SetRunningCost(p["Q"], p["R"])
    ilqr.SetTerminalCost(p["Qf"])
    ilqr.SetInitialGuess(W.arm27c_u_guess(p["N"]))
    states, inputs, solve_time, optimal_cost = ilqr.Solve()
    assert states.shape == (27, 50) and inputs.shape == (7, 49) and ilqr.K.shape == (7, 27, 49) and ilqr.kappa.shape == (7, 49)
    g, _ = load_golden("arm27c_solve_0")
    assert abs(optimal_cost - g["L"]) < 1e-7 * g["L"]                 # (central differences here, exact Jacobians in the fixture)
    f = tmp_path / "side.npz"
    ilqr.SaveSolution(str(f))
    z = np.load(f)
    assert sorted(z.files) == ["K", "t", "u_bar", "x_bar"] and z["x_bar"].shape == (27, 49) and z["K"].shape == (7, 27, 49)
