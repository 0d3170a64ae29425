"""Pins the C restatement (oracle/ilqr_oracle.c — the reported CPU baseline) against the
NumPy oracle (same FD Jacobians; ~1e-9) and against the reference's goldens (exact Jacobians
there, so FD tolerance).  CPU only."""
import numpy as np
import pytest

from common import load_golden, make_oracle, rel_err


@pytest.mark.parametrize("name", ["pendulum_c1", "pendulum_c2_01", "pendulum_kp_setinterval5", "acrobot_kp_adaptivejerk"])
def test_c_oracle_matches_numpy_oracle(name):
    from oracle import c_oracle, models_np as M
    g, prob = load_golden(name)
    minN = 5 if "setinterval5" in name else 1
    kp = ("setInterval", minN, 0, 0.0, 0.0)
    o = make_oracle(prob, keypoint=kp, jacobian="fd", fd_step=1e-5)
    o.set_problem(g["x0"], prob["x_nom"], prob["Q"], prob["R"], prob["Qf"], g["u_guess"])
    x, u, L, hist = o.solve()
    r = c_oracle.solve_batch(M.Model(prob["model_id"], prob["dt"]), prob, g["x0"][None], g["u_guess"], minN=minN)
    assert r["iters"][0] == len(hist) and r["ls"][0] == sum(h[2] for h in hist) and r["status"][0] == 0
    assert abs(r["cost"][0] - L) < 1e-10 * abs(L)
    # the acrobot optimum is flat (cost agrees to 1e-10 while x moves by 1e-7): FD round-off level
    tx = 1e-6 if prob["model_id"] == 1 else 1e-9
    assert rel_err(r["x_bar"][0], x) < tx and rel_err(r["u_bar"][0], u) < 10 * tx
    assert rel_err(r["K"][0], o.K) < 100 * tx
    # kappa -> 0 at the optimum: what is left is finite-difference noise, compare absolutely
    assert np.max(np.abs(r["kappa"][0] - o.kappa)) < (1e-4 if prob["model_id"] == 1 else 1e-7)


def test_c_oracle_vs_reference_golden_and_threads():
    from oracle import c_oracle, models_np as M
    names = [f"pendulum_c2_{i:02d}" for i in range(8)]
    gs = [load_golden(n) for n in names]
    prob = gs[0][1]
    x0 = np.stack([g["x0"] for g, _ in gs])
    r1 = c_oracle.solve_batch(M.Model(prob["model_id"], prob["dt"]), prob, x0, np.zeros((1, prob["N"] - 1)), nthreads=1)
    r4 = c_oracle.solve_batch(M.Model(prob["model_id"], prob["dt"]), prob, x0, np.zeros((1, prob["N"] - 1)), nthreads=4)
    assert r1["threads"] == 1 and r4["threads"] == 4
    assert np.array_equal(r1["cost"], r4["cost"]) and np.array_equal(r1["x_bar"], r4["x_bar"])   # threading is bitwise neutral
    for b, (g, _) in enumerate(gs):
        assert r1["iters"][b] == len(g["hist"])
        assert abs(r1["cost"][b] - g["L"]) < 1e-8 * abs(g["L"])
        assert np.max(np.abs(r1["x_bar"][b] - g["x_bar"])) < 1e-6
        assert rel_err(r1["K"][b], g["K"]) < 1e-5


@pytest.mark.parametrize("name", ["acrobot_mpc_0", "acrobot_mpc_1", "synth36_mpc_0", "quad_mpc_0", "quad3d_mpc_0", "quad3d_mpc_1", "arm27_mpc_0", "arm27c_mpc_0"])
def test_c_oracle_mpc_loop_vs_reference_golden(name):
    """oracle_mpc_batch (shift warm start, moving target, gains persisting across solves - SURVEY F10)
    against MPC sequences recorded from the unmodified reference (exact Jacobians there, central FD here:
    FD tolerances; iteration counts exact)."""
    from oracle import c_oracle, models_np as M, problems as P
    g, prob = load_golden(name)
    n, N = g["xs"].shape[1], prob["N"]
    m = g["us"].shape[1]
    replan, R = int(g["replan"]), len(g["Ls"]) - 1
    step = None
    ug = np.zeros((m, N - 1))
    if "move_target" in g:
        step = np.zeros(n)
        step[int(g["move_target"][0])] = g["move_target"][1]
    if prob["model_id"] >= 4:
        ug = {4: P.synth36_u_guess, 5: P.planar_quad_u_guess, 6: P.quad3d_u_guess, 7: P.arm27_u_guess, 8: P.arm27c_u_guess}[prob["model_id"]](N)
    r = c_oracle.mpc_batch(M.Model(prob["model_id"], prob["dt"]), prob, g["x0"][None], ug, R, replan, target_step=step)
    assert r["status"][0] == 0
    # (arm + ball: the hand-ball contact's curvature k / sigma^2 = 6e7 on a 0.2 kg ball makes the h^2 truncation term of the
    #  central differences visible in the cost: 1.0e-7 relative observed against the reference's exact Jacobians)
    tolL = 5e-7 if prob["model_id"] in (7, 8) else 1e-8
    assert int(r["first"][0, 1]) == g["iters"][0] and abs(r["first"][0, 0] - g["Ls"][0]) < tolL * abs(g["Ls"][0])
    log = r["log"][0]
    assert np.array_equal(log[:, -1].astype(int), g["iters"][1:])
    assert rel_err(log[:, -2], g["Ls"][1:]) < tolL
    for k in range(R):
        want = g["xs"][k + 1][:, 0]                # (the acrobot optimum is flat: x moves at FD-noise level, 1e-6 relative)
        assert np.max(np.abs(log[k, :n] - want)) < 1e-6 * max(1.0, np.max(np.abs(want)))
    assert rel_err(r["x_bar"][0], g["xs"][-1]) < 1e-6 and rel_err(r["u_bar"][0], g["us"][-1]) < 1e-5
    assert rel_err(r["K"][0], g["Ks"][-1]) < 1e-5


@pytest.mark.parametrize("name", ["quad_solve_0", "quad_infeasible_0", "quad_infeasible_1",
                                  "quad3d_solve_0", "quad3d_solve_1", "quad3d_infeasible_0", "arm27_solve_0", "arm27_solve_1",
                                  "arm27c_solve_0", "arm27c_solve_1"])
def test_c_oracle_planar_quadruped_vs_reference_golden(name):
    """The C restatement of the articulated-body model, of the 3-D quadruped, of the arm + ball and of the infeasible-step rule
    (L = inf, ilqr.py:315-323) against solves recorded from the reference (exact Jacobians there, central FD here)."""
    from oracle import c_oracle, models_np as M
    g, prob = load_golden(name)
    r = c_oracle.solve_batch(M.Model(prob["model_id"], prob["dt"], prob.get("params")), prob, g["x0"][None], g["u_guess"])
    assert r["status"][0] == 0 and r["iters"][0] == len(g["hist"]) and r["ls"][0] == int(g["hist"][:, 2].sum())
    assert abs(r["cost"][0] - g["L"]) < 1e-7 * abs(g["L"])
    assert np.max(np.abs(r["x_bar"][0] - g["x_bar"])) < 1e-5 and rel_err(r["K"][0], g["K"]) < 1e-4


@pytest.mark.parametrize("name", ["pendulum_kp_setinterval5", "pendulum_kp_adaptivejerk", "pendulum_kp_iterativeerror",
                                  "acrobot_kp_adaptivejerk", "acrobot_kp_iterativeerror",
                                  "arm27_kp_adaptivejerk", "arm27_kp_iterativeerror", "arm27c_kp_adaptivejerk", "arm27c_kp_iterativeerror"])
def test_c_oracle_keypoint_methods_vs_numpy_oracle_and_golden(name):
    """setInterval / adaptiveJerk / iterativeError (ilqr.py:417-593) in the C restatement: against the NumPy oracle with
    the same central differences (per-iteration trials, step sizes and key-point counts exact, the last key-point
    list exact), and against the list the unmodified reference recorded (exact Jacobians there)."""
    from common import golden_keypoint
    from oracle import c_oracle, models_np as M
    g, prob = load_golden(name)
    kp = golden_keypoint(g)
    o = make_oracle(prob, keypoint=kp, jacobian="fd", fd_step=1e-5)
    o.set_problem(g["x0"], prob["x_nom"], prob["Q"], prob["R"], prob["Qf"], g["u_guess"])
    x, u, L, hist = o.solve()
    hist = np.array(hist)
    r = c_oracle.solve_batch(M.Model(prob["model_id"], prob["dt"]), prob, g["x0"][None], g["u_guess"], keypoint=kp, hist_cap=64)
    it = int(r["iters"][0])
    assert it == len(hist) and r["status"][0] == 0
    h = r["hist"][0][:it]
    assert np.array_equal(h[:, 1:3], hist[:, 1:3])                                        # eps and trials of every iteration
    assert np.array_equal(h[:, 3], np.round(hist[:, 3] * (prob["N"] - 1) / 100.0))        # key-point count of every iteration
    assert rel_err(h[:, 0], hist[:, 0]) < 1e-8          # (central differences of the acrobot: round-off level 1e-9)
    nk = int(r["kp_count"][0])
    assert np.array_equal(r["kp_list"][0][:nk], o.keypoints)
    assert np.array_equal(r["kp_list"][0][:nk], g["kp_last"]) and it == len(g["hist"])
    assert rel_err(r["x_bar"][0], x) < 1e-6 and abs(r["cost"][0] - L) < 1e-8 * abs(L)
