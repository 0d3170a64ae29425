"""GPU parity tests added in round 2 (all through the C ABI):
  * the benchmarked C3 / C5 code paths at their full sizes (mi_ilqr_mpc_run, every problem, every re-solve)
    against the C oracle's receding-horizon loop (pinned to the reference's MPC goldens by tests/test_c_oracle.py);
  * C4 (stiff contact, central differences): lockstep against the oracle with FD on both sides, and the
    end-to-end deviation held to the problem's own sensitivity (oracle vs oracle with x0 moved by one ulp);
  * cost matrices outside the symmetric-PSD class, Reset() semantics, per-re-solve status, the statistics
    epilogue under load, the device-side layout conversions of mi_ilqr_get/_set.
"""
import os

import numpy as np
import pytest

from common import load_golden, make_oracle, rel_err, assert_flip_budget
from test_gpu_parity import make_solver

pytestmark = pytest.mark.gpu


# ----------------------------------------------------------------------------------
# the benchmarked MPC paths at BASELINE.json's sizes
# ----------------------------------------------------------------------------------
def _check_mpc_against_oracle(s, r, first_it, first_L, n, tol_L, tol_x, budget="c3_mpc_full"):
    log = s.mpc_log
    assert np.array_equal(first_it, r["first"][:, 1].astype(int))
    assert np.max(np.abs(first_L - r["first"][:, 0]) / np.abs(r["first"][:, 0])) < tol_L
    assert (s.status == 0).all() and (r["status"] == 0).all()
    same = log[:, :, -1] == r["log"][:, :, -1]                        # iterations of every re-solve of every problem
    full = same.all(axis=1)
    # a flipped line-search decision at round-off level: as many problems as were observed to have one (common.FLIP_BUDGET) ...
    assert_flip_budget(budget, full)
    relL = np.abs(log[:, :, -2] - r["log"][:, :, -2]) / np.abs(r["log"][:, :, -2])
    assert np.max(relL[full]) < tol_L
    # ... and such a problem must still track the oracle's closed loop: cost of every re-solve within 1e-3
    assert np.all(relL[~full] < 1e-3)
    scale = max(1.0, np.max(np.abs(r["log"][:, :, :n])))
    assert np.max(np.abs(log[full][:, :, :n] - r["log"][full][:, :, :n])) < tol_x * scale     # x0 of every re-solve
    assert np.max(np.abs(s.x_bar[full] - r["x_bar"][full])) < tol_x * scale
    assert rel_err(s.K[full], r["K"][full]) < 1e-5
    assert np.array_equal(s.ls_trials[full], r["ls"][full])            # line-search trials, summed over the 1 + R solves
    return log


def test_c3_full_size_mpc_run_vs_oracle():
    """C3 as benchmarked: acrobot n=4 m=1 N=40, B=512, cold solve + MPCRun(50, 2) (ONE launch, state in LDS
    between re-solves) against oracle_mpc_batch on all 512 problems (acrobot.py:131-162)."""
    from drake_ddp_amd import workloads as W
    from oracle import c_oracle, models_np as M
    a = W.acrobot_problem()
    B = 512
    x0 = W.acrobot_batch_x0(B)
    s = make_solver(a, B=B, jac="fd")
    s.SetInitialState(x0)
    s.SetInitialGuess(np.zeros((1, a["N"] - 1)))
    s.Solve()
    first_it, first_L, ls0 = s.iterations.copy(), s.cost.copy(), s.ls_trials.copy()
    st = s.MPCRun(50, 2)
    r = c_oracle.mpc_batch(M.Model(a["model_id"], a["dt"]), a, x0, np.zeros((1, a["N"] - 1)), 50, 2)
    r["ls"] = r["ls"] - ls0                                            # the device counter restarts with the MPC launch
    log = _check_mpc_against_oracle(s, r, first_it, first_L, 4, tol_L=5e-8, tol_x=1e-5)
    assert st.total_iters == int(log[:, :, -1].sum()) and st.n_converged == B
    assert np.array_equal(s.iterations, log[:, :, -1].sum(axis=1).astype(int))


def test_c5_full_size_mpc_run_vs_oracle():
    """C5 as benchmarked: n=36 m=12 N=40, B=64, cold solve + MPCRun(100, 4, moving target) in one launch of the
    workgroup-per-problem kernel, against oracle_mpc_batch on all 64 problems (mini_cheetah.py:186-213)."""
    from drake_ddp_amd import workloads as W
    from oracle import c_oracle, models_np as M
    q = W.synth36_problem()
    B = 64
    x0, ug = W.synth36_batch_x0(B), W.synth36_u_guess(q["N"])
    step = np.zeros(36)
    step[0] = W.SYNTH_TARGET_VEL * q["dt"] * 4
    s = make_solver(q, B=B, jac="fd")
    s.SetInitialState(x0)
    s.SetInitialGuess(ug)
    s.Solve()
    first_it, first_L, ls0 = s.iterations.copy(), s.cost.copy(), s.ls_trials.copy()
    st = s.MPCRun(100, 4, target_step=step)
    r = c_oracle.mpc_batch(M.Model(q["model_id"], q["dt"]), q, x0, ug, 100, 4, target_step=step)
    r["ls"] = r["ls"] - ls0
    log = _check_mpc_against_oracle(s, r, first_it, first_L, 36, tol_L=5e-8, tol_x=1e-6, budget="c5_mpc_full")
    assert st.total_iters == int(log[:, :, -1].sum()) and st.n_converged == B
    assert np.allclose(s.x_nom, q["x_nom"] + 100 * step)              # the handle's target followed the loop


# ----------------------------------------------------------------------------------
# C4: central differences on the stiff contact model
# ----------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["cartpole_wall_c4_0", "cartpole_wall_c4_1"])
def test_c4_lockstep_vs_oracle_with_finite_differences(name):
    """test_lockstep_vs_oracle with jac = fd on BOTH sides (BASELINE's C4 mandates FD; same h): every
    iteration restarted from the device's own state.  The rollout agrees to round-off; Jacobians to the
    round-off of a central difference (1e-16 / 2h relative to |f| ~ 1, i.e. ~1e-10 absolute on entries as small
    as 1e-3); the gains inherit that through Quu^-1, and kappa / dV_coeff - which vanish at the optimum -
    are compared relative to their largest entry of the iteration."""
    g, prob = load_golden(name)
    s = make_solver(prob, jac="fd")
    o = make_oracle(prob, jacobian="fd", fd_step=1e-5)
    n, m, N = g["x_bar"].shape[0], g["u_bar"].shape[0], prob["N"]
    st = dict(x_bar=np.zeros((n, N)), u_bar=np.array(g["u_guess"], float).reshape(m, N - 1),
              K=np.zeros((m, n, N - 1)), kappa=np.zeros((m, N - 1)), dV_coeff=np.zeros(N - 1))
    L = np.inf
    s.SetInitialState(g["x0"][None])
    iters = min(len(g["hist"]), 40)
    for it in range(iters):
        s.set_state(**{k: v[None] for k, v in st.items()})
        Lg, eps, ls = s.stage_forward(L)
        s.stage_backward()
        o.set_problem(g["x0"], prob["x_nom"], prob["Q"], prob["R"], prob["Qf"], st["u_bar"])
        o.x_bar, o.K, o.kappa, o.dV = st["x_bar"].copy(), st["K"].copy(), st["kappa"].copy(), st["dV_coeff"].copy()
        Lo, eps_o, ls_o = o.forward(L)
        o.backward()
        assert ls[0] == ls_o and eps[0] == eps_o, (it, ls[0], ls_o)
        assert abs(Lg[0] - Lo) < 1e-12 * abs(Lo), (it, Lg[0], Lo)
        assert rel_err(s.x_bar[0], o.x_bar) < 1e-12 and rel_err(s.u_bar[0], o.u_bar) < 1e-11, it
        assert rel_err(s.fx[0], o.fx) < 1e-9 and rel_err(s.fu[0], o.fu) < 5e-7, it
        assert rel_err(s.K[0], o.K) < 1e-5, it
        assert rel_err(s.kappa[0], o.kappa) < 5e-4 and rel_err(s.dV_coeff[0], o.dV) < 5e-4, it
        st = dict(x_bar=s.x_bar[0], u_bar=s.u_bar[0], K=s.K[0], kappa=s.kappa[0], dV_coeff=s.dV_coeff[0])
        L = Lg[0]


@pytest.mark.parametrize("name", ["cartpole_wall_c4_0", "cartpole_wall_c4_1"])
def test_c4_end_to_end_deviation_is_the_problems_own_sensitivity(name):
    """What "same basin" means for C4, as a number.  The oracle (central FD, h = 1e-5) solves the golden's problem
    three times: as given, and with the pole angle of x0 moved by ONE ulp up / down.  The device (same FD)
    solves it once.  All four runs must take the same line-search decisions in every iteration and the same
    number of iterations, and the device's cost history may deviate from the unperturbed oracle's by no more
    than 10x what a one-ulp change of the input does to the oracle itself (errors of 1e-16 are amplified to
    1e-7..1e-5 within a few iterations of this stiff contact problem on ANY implementation)."""
    g, prob = load_golden(name)
    s = make_solver(prob, jac="fd", single=True, hist_cap=256)
    s.SetInitialState(g["x0"])
    s.SetInitialGuess(g["u_guess"])
    x, u, _, L = s.Solve()
    it = int(s.iterations[0])
    h = s.history[0][:it]
    runs = []
    for k in range(3):
        o = make_oracle(prob, jacobian="fd", fd_step=1e-5)
        x0 = np.array(g["x0"], float)
        if k:
            x0[1] = np.nextafter(x0[1], np.inf if k == 1 else -np.inf)
        o.set_problem(x0, prob["x_nom"], prob["Q"], prob["R"], prob["Qf"], g["u_guess"])
        xo, uo, Lo, hist = o.solve()
        runs.append((np.array(hist), xo, uo))
    A = runs[0][0]
    assert it == len(A) == len(runs[1][0]) == len(runs[2][0]) == len(g["hist"])      # (the AD golden takes as many)
    for other in (h, runs[1][0], runs[2][0]):
        assert np.array_equal(other[:, 1:3], A[:, 1:3])                              # eps and trial count of every iteration
    d_hip = np.abs(h[:, 0] - A[:, 0]) / np.abs(A[:, 0])
    d_ulp = np.maximum(np.abs(runs[1][0][:, 0] - A[:, 0]), np.abs(runs[2][0][:, 0] - A[:, 0])) / np.abs(A[:, 0])
    # running maximum: the perturbation's effect on one iteration's cost can pass through zero
    envelope = np.maximum.accumulate(d_ulp)
    assert np.all(d_hip <= 10.0 * envelope + 1e-13), (d_hip, envelope)
    assert abs(L - A[-1, 0]) <= 10.0 * envelope[-1] * abs(A[-1, 0]) + 1e-12
    dx_ulp = max(np.max(np.abs(runs[1][1] - runs[0][1])), np.max(np.abs(runs[2][1] - runs[0][1])))
    assert np.max(np.abs(x - runs[0][1])) <= 10.0 * dx_ulp + 1e-12


# ----------------------------------------------------------------------------------
# cost matrices the fast backward passes do not cover
# ----------------------------------------------------------------------------------
@pytest.mark.parametrize("cfg", ["pendulum", "acrobot", "acrobot_long"])
def test_asymmetric_and_indefinite_costs_follow_the_reference(cfg):
    """The reference accepts any Q/Qf (lxx = 2Q, never symmetrized: ilqr.py:182,653-667).  The scan / MFMA backward
    passes assume Vxx = Vxx^T and PSD terms; mi_ilqr_set_cost detects other matrices and the kernels run the
    reference's recursion verbatim: one backward pass and a short solve against the NumPy oracle."""
    from drake_ddp_amd import workloads as W
    rng = np.random.default_rng(11)
    prob = dict(W.pendulum_problem() if cfg == "pendulum" else W.acrobot_problem(N=150 if cfg == "acrobot_long" else 40))
    n = prob["Q"].shape[0]
    prob["Q"] = prob["Q"] + prob["dt"] * 0.02 * np.triu(rng.uniform(0.5, 1.0, (n, n)), 1)       # asymmetric
    prob["Qf"] = prob["Qf"] + np.tril(rng.uniform(1.0, 3.0, (n, n)), -1)                        # asymmetric
    prob["delta"] = 1e-3
    x0 = (W.pendulum_batch_x0(8, seed=5) if n == 2 else W.acrobot_batch_x0(8, seed=5))[:3]
    N = prob["N"]
    ug = 0.1 * rng.standard_normal((3, 1, N - 1))
    s = make_solver(prob, B=3, jac="ad", max_iters=6)
    s.SetInitialState(x0)
    s.SetInitialGuess(ug)
    s.stage_forward(np.inf)
    s.stage_backward()
    K1, k1, dV1, fx1 = s.K.copy(), s.kappa.copy(), s.dV_coeff.copy(), s.fx.copy()
    s.Reset()                                                          # (the stage calls left gains and a trajectory behind)
    s.SetInitialGuess(ug)
    x, u, _, L = s.Solve()
    from oracle.ilqr_np import LinesearchFailed
    n_eps = int(np.floor(np.log(1e-8) / np.log(prob["beta"]))) + 1      # step sizes >= 1e-8 (ilqr.py:302)
    outcomes = set()
    for b in range(3):
        o = make_oracle(prob)
        o.max_iters = 6
        o.set_problem(x0[b], prob["x_nom"], prob["Q"], prob["R"], prob["Qf"], ug[b])
        o.forward(np.inf)
        o.backward()
        assert rel_err(fx1[b], o.fx) < 1e-11
        assert rel_err(K1[b], o.K) < 1e-9 and rel_err(k1[b], o.kappa) < 1e-9 and rel_err(dV1[b], o.dV) < 1e-9
        # the Solve loop of the oracle (ilqr.py:692-708), iteration by iteration: with an asymmetric Q the reference's
        # lx = 2Qx - 2 x_nom^T Q is not the gradient of its own cost, so its line search may run out of step sizes
        # (RuntimeError, ilqr.py:337) - the device must then report exactly that, after the same iterations
        o = make_oracle(prob)
        o.set_problem(x0[b], prob["x_nom"], prob["Q"], prob["R"], prob["Qf"], ug[b])
        Lo, done, failed, trials = np.inf, 0, False, 0
        while done < 6:
            try:
                L_new, eps, ls = o.forward(Lo)
            except LinesearchFailed:
                failed = True
                break
            o.backward()
            trials += ls
            done += 1
            improvement, Lo = Lo - L_new, L_new
            if not improvement > prob["delta"]:
                break
        assert s.iterations[b] == done, (b, s.iterations[b], done)
        if failed:
            assert s.status[b] == 2 and s.ls_trials[b] == trials + n_eps
        else:
            assert s.ls_trials[b] == trials and abs(L[b] - Lo) < 1e-9 * abs(Lo)
        if done:
            # (gains of up to 2e4 at the horizon's end, six iterations of accumulated round-off: 1e-5 relative)
            assert rel_err(x[b], o.x_bar) < 1e-8 and rel_err(s.K[b], o.K) < 1e-5
        outcomes.add(failed)
    assert len(outcomes) == 2 or cfg != "acrobot_long"                # both outcomes are exercised on the long horizon
    # the symmetric-PSD case is unaffected: same handle, regular matrices again -> the fast passes
    base = dict(W.pendulum_problem() if cfg == "pendulum" else W.acrobot_problem(N=N))
    s.SetRunningCost(base["Q"], base["R"])
    s.SetTerminalCost(base["Qf"])
    s.Reset()
    s.SetInitialGuess(ug)
    s.Solve()
    assert np.all(np.isfinite(s.cost))


@pytest.mark.parametrize("name", ["synth36_stage", "quad3d_stage"])
def test_large_path_follows_asymmetric_costs(name):
    """Until round 6 the n = 36 / 37 kernels refused cost matrices that are not symmetric (this test asserted the refusal); the
    reference takes any (ilqr.py:130-146) and never symmetrizes (:180-184, :651-667).  Stage level on the reference-generated
    fixtures' trajectories and Jacobians with Q, R AND Qf made asymmetric: K, kappa, dV against oracle.backward() and against the
    reference's recursion in extended precision (the device no further from it than 20 x the fp64 NumPy pass is, or 1e-10) -
    and NOT what the symmetrized matrices give; with symmetric matrices the same handle is bitwise the matrix-core pass."""
    from common import backward_errors
    g, prob = load_golden(name)
    rng = np.random.default_rng(5)

    def asym(A, upper, floor):
        d = np.sqrt(np.abs(np.diag(A)) + floor)
        T = rng.uniform(0.5, 1.0, A.shape)
        return A + 0.1 * (d[:, None] * (np.triu(T, 1) if upper else np.tril(T, -1)) * d[None, :])
    Q, R, Qf = asym(prob["Q"], True, 1e-3 * prob["dt"]), asym(prob["R"], False, 0.0), asym(prob["Qf"], False, 1e-3)
    o = make_oracle(prob)
    o.set_problem(g["x0"], prob["x_nom"], Q, R, Qf, g["pre_u_bar"])
    o.x_bar, o.u_bar, o.fx, o.fu = g["roll_x"], g["roll_u"], g["fx"], g["fu"]
    o.backward()
    s = make_solver(prob, jac="ad", on_indefinite="continue")
    s.SetRunningCost(Q, R); s.SetTerminalCost(Qf)
    s.SetInitialState(g["x0"][None])                         # (no SetInitialGuess: u_bar is set as state below)
    s.set_state(x_bar=g["roll_x"][None], u_bar=g["roll_u"][None], fx=g["fx"][None], fu=g["fu"][None])
    s.stage_backward()
    dev = (s.K[0], s.kappa[0], s.dV_coeff[0])
    e_dev, e_ref, cond = backward_errors(dev, o)
    per = [rel_err(a_, b_) for a_, b_ in zip(dev, (o.K, o.kappa, o.dV))]
    print(f"{name}: device vs NumPy K {per[0]:.1e} kappa {per[1]:.1e} dV {per[2]:.1e}; vs extended precision {e_dev:.1e} (NumPy {e_ref:.1e}), cond(Quu) {cond:.1e}")
    # (the 3-D quadruped's fixture with these matrices: cond(Quu) 3e10 - the fp64 reference itself is 2e-5 from the extended pass,
    #  the device 1.4e-5; the 36-state chain: both at 1e-12)
    assert e_ref < 1e-4 and e_dev < max(1e-10, 20 * e_ref), (e_dev, e_ref, cond)
    osym = make_oracle(prob)
    osym.set_problem(g["x0"], prob["x_nom"], 0.5 * (Q + Q.T), 0.5 * (R + R.T), 0.5 * (Qf + Qf.T), g["pre_u_bar"])
    osym.x_bar, osym.u_bar, osym.fx, osym.fu = g["roll_x"], g["roll_u"], g["fx"], g["fu"]
    osym.backward()
    assert rel_err(osym.K, o.K) > 1e-6                       # (the asymmetric parts are seen)
    # symmetric matrices on the SAME handle: the matrix-core pass, bitwise what a fresh handle returns
    s.SetRunningCost(prob["Q"], prob["R"]); s.SetTerminalCost(prob["Qf"])
    s.stage_backward()
    f = make_solver(prob, jac="ad", on_indefinite="continue")
    f.SetInitialState(g["x0"][None])
    f.set_state(x_bar=g["roll_x"][None], u_bar=g["roll_u"][None], fx=g["fx"][None], fu=g["fu"][None])
    f.stage_backward()
    assert np.array_equal(s.K, f.K) and np.array_equal(s.kappa, f.kappa) and rel_err(s.K[0], g["post_K"]) < 1e-9


# ----------------------------------------------------------------------------------
# Reset(), per-re-solve status
# ----------------------------------------------------------------------------------
@pytest.mark.parametrize("cfg", ["pendulum", "synth36"])
def test_reset_is_a_freshly_constructed_solver(cfg):
    """mi_ilqr_reset = a new reference object (ilqr.py:70-83): ALL persistent state zero, including u_bar - a
    Solve() after Reset() without SetInitialGuess starts from u_bar = 0, not from the previous solution."""
    from drake_ddp_amd import workloads as W
    if cfg == "pendulum":
        prob, x0 = W.pendulum_problem(), W.pendulum_batch_x0(4, seed=2)
        ug = 0.3 * np.ones((1, prob["N"] - 1))
    else:
        prob, x0 = W.synth36_problem(), W.synth36_batch_x0(4)
        ug = W.synth36_u_guess(prob["N"])
    s = make_solver(prob, B=4, jac="fd")
    s.SetInitialState(x0)
    s.SetInitialGuess(ug)
    s.Solve()
    assert np.abs(s.u_bar).max() > 0
    s.Reset()
    assert np.abs(s.u_bar).max() == 0 and np.abs(s.x_bar).max() == 0 and np.abs(s.K).max() == 0
    xa, ua, _, La = s.Solve()
    fresh = make_solver(prob, B=4, jac="fd")
    fresh.SetInitialState(x0)
    xb, ub, _, Lb = fresh.Solve()                                       # never given a guess: u_bar = 0 (ilqr.py:71)
    assert np.array_equal(La, Lb) and np.array_equal(xa, xb) and np.array_equal(ua, ub)
    assert np.array_equal(s.iterations, fresh.iterations)


def test_mpc_status_is_that_of_the_last_resolve():
    """A re-solve that hits the iteration cap must not mark the re-solves after it: the single-launch loop
    leaves the status a host loop of shift + Solve() calls leaves (each solve overwrites it)."""
    from drake_ddp_amd import workloads as W
    a = W.acrobot_problem()
    B, R = 64, 20
    x0 = W.acrobot_batch_x0(B)
    twins = []
    for _ in range(2):
        s = make_solver(a, B=B, jac="fd", max_iters=3)
        s.SetInitialState(x0)
        s.SetInitialGuess(np.zeros((1, a["N"] - 1)))
        s.Solve()
        twins.append(s)
    dev, host = twins
    dev.MPCRun(R, 2)
    seen_capped = np.zeros(B, bool)
    for r in range(R):
        host.MPCShift(2)
        host.solve_resident()
        seen_capped |= host.status == 1
    assert np.array_equal(dev.mpc_log[:, -1, -1].astype(int), host.iterations)
    assert np.array_equal(dev.status, host.status)
    assert (seen_capped & (host.status == 0)).any()                   # capped once, converged later: not sticky


# ----------------------------------------------------------------------------------
# statistics epilogue under load; layout conversions on the device
# ----------------------------------------------------------------------------------
def test_statistics_epilogue_under_load(tmp_path):
    """The in-kernel batch statistics (last workgroup reduces; results published with write-through stores,
    s_waitcnt vmcnt(0), then the ticket) at B = 1024 - every workgroup finishing close to the others -
    repeated 200 times, against the per-problem arrays of the same solve."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = f"""
import sys, numpy as np
sys.path.insert(0, {root!r}); sys.path.insert(0, {os.path.join(root, 'tests')!r})
from drake_ddp_amd import workloads as W
from test_gpu_parity import make_solver
prob = W.pendulum_problem()
x0 = W.pendulum_batch_x0(1024)
s = make_solver(prob, B=1024, jac='fd', max_iters=9, hist_cap=2)
s.SetInitialState(x0); s.SetInitialGuess(np.zeros((1, prob['N'] - 1)))
s.Solve()
it, status, cost, ls = s.iterations, s.status, s.cost, s.ls_trials
conv = np.where(status == 0)[0]
want = (int(it.sum()), int(ls.sum()), int((status == 0).sum()), int((status == 1).sum()), int(it.max()),
        int(conv[np.argmin(cost[conv])]), float(cost[conv].min()))
bad = 0
for rep in range(200):
    s.rearm(cold=True)
    st = s.solve_resident()
    got = (st.total_iters, st.total_ls_trials, st.n_converged, st.n_max_iters, st.max_iters_seen, st.best_index, st.best_cost)
    bad += got != want
print('BAD', bad, want)
sys.exit(1 if bad else 0)
"""
    for forced in ("0", "1"):                                           # in-kernel epilogue, then the separate kernel
        r = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, timeout=600,
                           env=dict(os.environ, MI_ILQR_STATS_KERNEL=forced))
        assert r.returncode == 0, (forced, r.stdout[-500:], r.stderr[-1500:])


@pytest.mark.parametrize("path", ["large", "throughput"])
def test_layout_conversions_round_trip(path):
    """mi_ilqr_set / mi_ilqr_get convert between the reference's time-last layout and the kernels' time-major
    (n = 36) or batch-minor (lane-per-problem) layouts ON THE DEVICE: set -> get is the identity for every
    trajectory field, at sizes that are not multiples of the 32 x 32 transpose tiles."""
    from drake_ddp_amd import workloads as W
    rng = np.random.default_rng(3)
    if path == "large":
        prob, B, kw = dict(W.synth36_problem(), N=23), 5, {}
    else:
        prob, B, kw = dict(W.acrobot_problem(), N=71), 77, dict(kernel_mode="throughput")
    s = make_solver(prob, B=B, jac="fd", **kw)
    n, m, N = s.n, s.m, s.N
    fields = dict(x_bar=(B, n, N), u_bar=(B, m, N - 1), K=(B, m, n, N - 1), kappa=(B, m, N - 1), dV_coeff=(B, N - 1),
                  fx=(B, n, n, N - 1), fu=(B, n, m, N - 1))
    vals = {k: rng.standard_normal(shp) for k, shp in fields.items()}
    s.set_state(**vals)
    for k, v in vals.items():
        assert np.array_equal(getattr(s, k), v), k
    # and the initial guess travels through the same conversion (SetInitialGuess -> u_bar)
    ug = rng.standard_normal((B, m, N - 1))
    s.SetInitialState(rng.standard_normal((B, n)) * 0.01)
    s.SetInitialGuess(ug)
    s._push_problem()
    assert np.array_equal(s.u_bar, ug)


def test_native_rccl_collective_single_rank():
    """mi_ilqr_comm_* / mi_ilqr_allreduce_min: the library's own RCCL communicator (what a C caller uses for the
    best-cost reduction).  One GPU here, so one rank: the id is created, the communicator initialized and real
    ncclAllReduce(min) calls run on the device, blocking and start/wait; argument errors are refused."""
    from drake_ddp_amd.dist import NativeComm
    from drake_ddp_amd._capi import MiIlqrError
    c = NativeComm(0, 1, 0)
    v = np.array([3.5, -1.25, 7.0])
    assert np.array_equal(c.allreduce_min(v), v)
    assert np.array_equal(c.start(v[:2]).wait(), v[:2])
    with pytest.raises(MiIlqrError):
        c.allreduce_min(np.zeros(65))                                  # MI_ILQR_COMM_MAX_COUNT
    # and together with a solve: the reduction of the batch's best cost
    from drake_ddp_amd import workloads as W
    prob = W.pendulum_problem()
    s = make_solver(prob, B=8, jac="fd")
    s.SetInitialState(W.pendulum_batch_x0(8))
    s.SetInitialGuess(np.zeros((1, prob["N"] - 1)))
    s.Solve()
    assert c.allreduce_min([s.stats.best_cost])[0] == s.cost.min()
    c2 = NativeComm.from_torch(0)                                      # no process group: a one-rank communicator
    assert c2.world == 1 and c2.allreduce_min([2.0])[0] == 2.0


# ----------------------------------------------------------------------------------
# (f)4: planar quadruped - articulated-body dynamics + ground contact on the workgroup-per-problem kernels,
# and the model that can declare a step infeasible (ilqr.py:315-323)
# ----------------------------------------------------------------------------------
@pytest.mark.parametrize("jac", ["ad", "fd"])
def test_quad_stage_level(jac):
    g, prob = load_golden("quad_stage")
    s = make_solver(prob, jac=jac)
    s.SetInitialState(g["x0"][None])
    s.SetInitialGuess(g["pre_u_bar"])
    s.set_state(x_bar=g["pre_x_bar"][None], K=g["pre_K"][None], kappa=g["pre_kappa"][None], dV_coeff=g["pre_dV"][None])
    x, u, L, ex = s.stage_rollout(1.0)                       # cooperative step: one lane per chain of the tree
    assert rel_err(x[0], g["roll_x"]) < 1e-10 and rel_err(u[0], g["roll_u"]) < 1e-10
    assert abs(L[0] - g["roll_L"]) < 1e-10 * abs(g["roll_L"])
    s.set_state(x_bar=x, u_bar=u)
    s.stage_linearize()                                      # whole-tree evaluation per (step, column) item
    tolj = 1e-10 if jac == "ad" else 1e-6                    # (contact curvature k/sigma^2 = 2.5e8: FD truncation ~1e-7 relative)
    assert rel_err(s.fx[0], g["fx"]) < tolj and rel_err(s.fu[0], g["fu"]) < tolj
    s.stage_backward()
    tolk = 1e-7 if jac == "ad" else 1e-4                     # (Quu is ill-conditioned: cond ~1e5 on this model)
    assert rel_err(s.K[0], g["post_K"]) < tolk
    assert rel_err(s.kappa[0], g["post_kappa"]) < (1e-7 if jac == "ad" else 1e-3)
    assert rel_err(s.dV_coeff[0], g["post_dV"]) < (1e-7 if jac == "ad" else 1e-3)


@pytest.mark.parametrize("name", ["quad_solve_0", "quad_infeasible_0", "quad_infeasible_1"])
def test_quad_solve_vs_reference_golden(name):
    """Whole solves recorded from the unmodified reference.  quad_infeasible_*: line-search trials are declared
    infeasible by the model (|v| bound 27.5), the reference turns them into L = inf and backs off - the
    device must take exactly the same step sizes."""
    g, prob = load_golden(name)
    s = make_solver(prob, jac="ad", single=True, hist_cap=32)
    s.SetInitialState(g["x0"])
    s.SetInitialGuess(g["u_guess"])
    x, u, _, L = s.Solve()
    iters = int(s.iterations[0])
    assert iters == len(g["hist"])
    h = s.history[0][:iters]
    assert np.array_equal(h[:, 1:3], g["hist"][:, 1:3])              # eps and trial count of every iteration
    assert rel_err(h[:, 0], g["hist"][:, 0]) < 1e-8 and abs(L - g["L"]) < 1e-8 * abs(g["L"])
    assert np.max(np.abs(x - g["x_bar"])) < 1e-7 and np.max(np.abs(u - g["u_bar"])) < 1e-6
    # (Jacobians at the FINAL trajectory, which agrees to ~1e-8: the contact curvature k/sigma^2 = 2.5e8 turns
    # that into ~1e-8 relative on fx)
    assert rel_err(s.K, g["K"]) < 1e-5 and rel_err(s.fx, g["fx"]) < 1e-6


def test_quad_batch_fd_vs_c_oracle_with_infeasible_trials():
    """64 seeded quadruped problems, central differences, default and tightened velocity bound, against the C
    oracle (pinned to the reference's quad goldens): iteration and line-search-trial counts of every problem,
    costs, trajectories; with the tight bound the counts must differ from the free run for some problems
    (the infeasibility rule is really exercised on the device)."""
    from drake_ddp_amd import workloads as W
    from oracle import c_oracle, models_np as M
    prob = W.planar_quad_problem()
    B = 64
    x0, ug = W.planar_quad_batch_x0(B), W.planar_quad_u_guess(prob["N"])
    res = {}
    for tag, vmax in (("free", 60.0), ("tight", 27.5)):
        par = np.array(M.DEFAULT_PARAMS[M.PLANAR_QUAD], float)
        par[8] = vmax
        p = dict(prob, params=par)
        s = make_solver(p, B=B, jac="fd")
        s.SetInitialState(x0)
        s.SetInitialGuess(ug)
        x, u, _, L = s.Solve()
        r = c_oracle.solve_batch(M.Model(p["model_id"], p["dt"], par), p, x0, ug)
        assert np.array_equal(s.status, r["status"])
        ok = s.status == 0
        same = ok & (s.iterations == r["iters"]) & (s.ls_trials == r["ls"])
        assert_flip_budget(f"quad_batch_{tag}_unconverged", ok)
        assert_flip_budget(f"quad_batch_{tag}", same[ok])
        rel = np.abs(L - r["cost"]) / np.abs(r["cost"])
        assert np.max(rel[same]) < 1e-6 and np.all(rel[ok & ~same] < 1e-2)
        assert np.max(np.abs(x[same] - r["x_bar"][same])) < 1e-4
        res[tag] = (s.iterations.copy(), s.ls_trials.copy())
    assert (res["free"][1] != res["tight"][1]).any()


@pytest.mark.parametrize("device_loop", [False, True])
def test_quad_mpc_moving_target(device_loop):
    from drake_ddp_amd.workloads import mpc_shift, planar_quad_u_guess
    g, prob = load_golden("quad_mpc_0")
    s = make_solver(prob, jac="ad")
    N, replan, R = prob["N"], int(g["replan"]), len(g["Ls"]) - 1
    s.SetInitialState(g["x0"][None])
    s.SetInitialGuess(planar_quad_u_guess(N))
    x, u, _, L = s.Solve()
    assert s.iterations[0] == g["iters"][0] and abs(L[0] - g["Ls"][0]) < 1e-8 * abs(g["Ls"][0])
    step = np.zeros(36)
    step[int(g["move_target"][0])] = g["move_target"][1]
    if device_loop:
        s.MPCRun(R, replan, target_step=step)
        log = s.mpc_log[0]
        assert np.array_equal(log[:, -1].astype(int), g["iters"][1:]) and rel_err(log[:, -2], g["Ls"][1:]) < 1e-8
    else:
        x_nom = prob["x_nom"].copy()
        for r in range(1, R + 1):
            x_nom = x_nom + step
            x0, ug = mpc_shift(x, u, replan)
            s.SetInitialState(x0); s.SetInitialGuess(ug); s.SetTargetState(x_nom)
            x, u, _, L = s.Solve()
            assert s.iterations[0] == g["iters"][r] and abs(L[0] - g["Ls"][r]) < 1e-8 * abs(g["Ls"][r])
    assert rel_err(s.x_bar[0], g["xs"][-1]) < 1e-7 and rel_err(s.K[0], g["Ks"][-1]) < 1e-5


def test_quad_full_size_mpc_run_vs_oracle():
    """The benchmarked planar-quadruped config: B=64, cold solve + MPCRun(100, 4, moving target) in one launch,
    every problem and re-solve against the C oracle's receding-horizon loop."""
    from drake_ddp_amd import workloads as W
    from oracle import c_oracle, models_np as M
    q = W.planar_quad_problem()
    B = 64
    x0, ug = W.planar_quad_batch_x0(B), W.planar_quad_u_guess(q["N"])
    step = np.zeros(36)
    step[0] = W.QUAD_TARGET_VEL * q["dt"] * 4
    s = make_solver(q, B=B, jac="fd")
    s.SetInitialState(x0)
    s.SetInitialGuess(ug)
    s.Solve()
    first_it, first_L, ls0 = s.iterations.copy(), s.cost.copy(), s.ls_trials.copy()
    st = s.MPCRun(100, 4, target_step=step)
    r = c_oracle.mpc_batch(M.Model(q["model_id"], q["dt"]), q, x0, ug, 100, 4, target_step=step)
    r["ls"] = r["ls"] - ls0
    log = _check_mpc_against_oracle(s, r, first_it, first_L, 36, tol_L=1e-6, tol_x=1e-5, budget="quad_mpc_full")
    assert st.n_converged == B and np.all(log[:, -1, 0] > 0.3)        # the trunk moved forward by 0.3 m or more


def test_per_iteration_stopwatches_and_console_table(capsys):
    """The reference times every iteration's line search, derivatives and backward pass (ilqr.py:364-372,696-702) and
    prints them per row; the device logs the same three spans per iteration in shader-clock cycles
    (MI_F_ITER_CYCLES): they add up to the solve's stage totals, and the single-problem Solve() prints one row per
    iteration with that iteration's own times (and keeps the last iteration's in time_fp / time_getDerivs /
    time_backwardsPass like the reference's attributes)."""
    from drake_ddp_amd import workloads as W
    from drake_ddp_amd.ilqr import IterativeLinearQuadraticRegulator
    from drake_ddp_amd.models import ModelSystem
    for prob, x0, ug in ((W.acrobot_problem(), W.acrobot_batch_x0(1)[0], np.zeros((1, 39))),
                         (W.synth36_problem(), W.synth36_batch_x0(1)[0], W.synth36_u_guess(40))):
        s = IterativeLinearQuadraticRegulator(ModelSystem(prob["model_id"], prob["dt"]), prob["N"], delta=prob["delta"],
                                              beta=prob["beta"], gamma=prob["gamma"], jacobian_mode="fd", verbose=True)
        s.SetTargetState(prob["x_nom"]); s.SetRunningCost(prob["Q"], prob["R"]); s.SetTerminalCost(prob["Qf"])
        s.SetInitialState(x0); s.SetInitialGuess(ug)
        s.Solve()
        it = int(s.iterations[0])
        ic = s.iteration_cycles[0][:it]
        tot = s.stage_cycles[0]
        assert np.all(ic[:, 3] > 0) and np.all(ic[:, 0] > 0) and np.all(ic[:, 2] > 0)
        assert abs(ic[:, 0].sum() - tot[0]) <= 1e-9 * tot[0] and abs(ic[:, 1].sum() - tot[1]) <= 1e-9 * max(tot[1], 1)
        assert abs(ic[:, 2].sum() - tot[2]) <= 1e-9 * tot[2] and ic[:, 3].sum() <= tot[3]
        assert np.all(ic[:, 3] >= ic[:, 0] + ic[:, 1] + ic[:, 2] - 1)
        out = capsys.readouterr().out
        rows = [l for l in out.splitlines() if l.strip() and l.strip()[0].isdigit()]
        assert len(rows) == it
        times = np.array([[float(v) for v in r.split()[4:]] for r in rows])      # derivs, pct, bp, fp, iter, cumulative
        assert np.all(np.diff(times[:, 5]) >= 0)                                  # cumulative time column
        assert it == 1 or len(set(ic[:, 3])) > 1                                  # real per-iteration times, not one average
        assert np.allclose(times[:, 4], ic[:, 3] * (s.stats.kernel_ms * 1e-3 / tot[3]), atol=6e-6)   # (5 printed decimals)
        assert s.time_backwardsPass > 0 and s.time_fp > 0


@pytest.mark.parametrize("cfg,B,forced", [("quad", 8, None), ("synth36", 8, None), ("quad", 64, "8"), ("quad", 3, "5"),
                                          ("quad3d", 8, None), ("quad3d", 64, None), ("quad3d", 5, "3"),
                                          ("arm27", 1, None), ("arm27", 48, None), ("arm27", 7, "3"), ("chainx", 5, "4"),
                                          ("chainx4", 6, "4"), ("chainx4", 3, "8")])
def test_cluster_linearization_is_bitwise_the_single_workgroup_one(cfg, B, forced, tmp_path):
    """With few problems per GPU the linearization of ONE problem is shared by a cluster of workgroups (leader +
    helpers, handshake through global memory, ilqr_large.hpp).  Every Jacobian entry is still computed by the
    same code on the same inputs, so everything the solve returns must be bitwise what a single workgroup per
    problem returns (MI_ILQR_CLUSTER=1) - including when the launch is oversubscribed (512 workgroups on 256 CUs:
    helpers that are not resident are never waited for) and for cluster sizes that do not divide anything.  Since round 4
    the dense (whole-step) linearization of the mid-size models is shared the same way: the arm + ball (by default up to
    B = 64) and a plugin chain (forced: plugins are not clustered by default - the library cannot know what their step costs).
    Round 6: `chainx4` is a (4, 1) chain, N = 24, whose step takes a few hundred cycles and whose whole trajectory, Jacobians and
    gains fit the vector L1 several times over - the shape on which round 5's hand-shake handed out stale data twice (its
    `buffer_inv sc0` drops nothing, tools/ubench/l1_probe.hip); early rounds now run on plugin models too, and with the cluster
    spread over XCDs (`order0`)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = f"""
import sys, numpy as np
sys.path.insert(0, {root!r}); sys.path.insert(0, {os.path.join(root, 'tests')!r})
from drake_ddp_amd import workloads as W
from test_gpu_parity import make_solver
if {cfg!r} == 'quad':
    prob, x0, ug = W.planar_quad_problem(), W.planar_quad_batch_x0(64)[:{B}], W.planar_quad_u_guess(40)
    step = np.zeros(36); step[0] = W.QUAD_TARGET_VEL * prob['dt'] * 4
elif {cfg!r} == 'quad3d':
    prob, x0, ug = W.quad3d_problem(target_vel=1.0), W.quad3d_batch_x0(64)[:{B}], W.quad3d_u_guess(40)
    step = np.zeros(37); step[4] = 1.0 * prob['dt'] * 4
elif {cfg!r} == 'arm27':
    prob, x0, ug = W.arm27_problem(), W.arm27_batch_x0(64)[:{B}], W.arm27_u_guess(50)
    step = np.zeros(27); step[12] = 0.002
elif {cfg!r}.startswith('chainx'):
    sys.path.insert(0, {os.path.join(root, 'examples', 'plugins')!r})
    import models as PM
    from drake_ddp_amd.ilqr import BatchedIterativeLQR
    nq, m, N, dt = (7, 7, 30, 0.02) if {cfg!r} == 'chainx' else (2, 1, 24, 0.02)
    n = 2 * nq
    step = np.zeros(n); step[0] = 0.01
else:
    prob, x0, ug = W.synth36_problem(), W.synth36_batch_x0(64)[:{B}], W.synth36_u_guess(40)
    step = np.zeros(36); step[0] = W.SYNTH_TARGET_VEL * prob['dt'] * 4
if {cfg!r}.startswith('chainx'):
    s = BatchedIterativeLQR(PM.build_chainx(nq, m, 0)(dt), N, {B}, delta=1e-3, beta=0.7, jacobian_mode='fd')
    s.SetTargetState(np.zeros(n)); s.SetRunningCost(dt * np.eye(n), dt * 0.05 * np.eye(m)); s.SetTerminalCost(5.0 * np.eye(n))
    x0, ug = 0.4 * np.random.default_rng(2).standard_normal(({B}, n)), np.zeros((m, N - 1))
else:
    s = make_solver(prob, B={B}, jac='fd')
s.SetInitialState(x0); s.SetInitialGuess(ug)
x, u, _, L = s.Solve()
fx0, it0, cs0 = s.fx.copy(), s.iterations.copy(), s.cluster_stats
s.MPCRun(6, 4, target_step=step)
cs = s.cluster_stats
np.savez(sys.argv[1], x=x, u=u, L=L, fx0=fx0, it0=it0, log=s.mpc_log, xm=s.x_bar, K=s.K, fx=s.fx, fu=s.fu, it=s.iterations, st=s.status, cs=cs, cs0=cs0)
"""
    # Round 5: the default launch places a cluster on ONE XCD and lets the helpers linearize the line search's first trial while the
    # leader is still rolling it out (early linearization); the variants switch that off, put the members in consecutive slots of
    # the XCD, or spread them over XCDs like rounds 2 - 4 did (round 6: early rounds and candidate groups open there too - the
    # hand-shake no longer depends on the placement).
    variants = [("cluster", {}), ("single", {"MI_ILQR_CLUSTER": "1"})]
    if (cfg, B) in (("quad", 8), ("quad3d", 64), ("arm27", 48), ("synth36", 8), ("quad3d", 5), ("chainx4", 6), ("chainx4", 3), ("chainx", 5)):
        variants += [("early0", {"MI_ILQR_EARLY": "0"}), ("order0", {"MI_ILQR_CLUSTER_ORDER": "0"}), ("order1", {"MI_ILQR_CLUSTER_ORDER": "1"})]
    if cfg == "arm27":       # mid-size kernels: the helpers also roll out line-search candidates 4 .. beside the leader's four (candidate groups)
        variants += [("groups0", {"MI_ILQR_LS_GROUPS": "0"})]
    outs = {}
    for tag, env in variants:
        env = dict(env)
        if forced is not None and tag != "single": env["MI_ILQR_CLUSTER"] = forced
        f = str(tmp_path / f"{tag}.npz")
        base = {k: v for k, v in os.environ.items() if not k.startswith("MI_ILQR_")}     # (a suite run with forced switches does not leak into the variants)
        r = subprocess.run([sys.executable, "-c", script, f], capture_output=True, text=True, timeout=300, env=dict(base, **env))
        assert r.returncode == 0, r.stderr[-2000:]
        outs[tag] = np.load(f)
    assert (outs["cluster"]["st"] == 0).all()
    for tag in outs:
        for k in outs["single"].files:
            if k not in ("cs", "cs0"): assert np.array_equal(outs[tag][k], outs["single"][k]), (tag, k)
    # columns of cluster_stats: helpers, regular rounds, rounds with every helper on the leader's XCD, early rounds opened, ... accepted,
    # candidate-group rounds
    cs = outs["cluster"]["cs"]
    assert (outs["single"]["cs"] == 0).all()
    resident = cs[:, 0] > 0                                  # (oversubscribed launches: a problem's helpers may never have been resident)
    if resident.any():
        c = cs[resident]
        assert (c[:, 2] == c[:, 1] + c[:, 4]).all()          # every round that used the helpers' Jacobians found the whole cluster on one XCD (a speed matter only, since round 6)
        assert c[:, 3].sum() > 0 and c[:, 4].sum() > 0.5 * c[:, 3].sum()      # early rounds ran - on the plugin chains too - and mostly hit
        print(cfg, B, "helpers", c[:, 0].min(), "-", c[:, 0].max(), "regular rounds", c[:, 1].sum(), "early opened / accepted", c[:, 3].sum(), c[:, 4].sum())
    if cfg == "arm27":
        # candidate groups: in the cold solve (48 problems: some backtrack) - not in the receding-horizon loop, whose target MOVES
        # (the helpers keep their own LDS copy of x_nom); and they can be switched off
        assert (outs["groups0"]["cs0"][:, 5] == 0).all() and (cs[:, 5] == 0).all()
        if B == 48: assert outs["cluster"]["cs0"][:, 5].sum() > 0
        print("candidate-group rounds of the cold solve", outs["cluster"]["cs0"][:, 5].sum())
    if "order0" in outs:
        o0 = outs["order0"]["cs"]
        assert (o0[:, 2] == 0).all() and o0[:, 3].sum() > 0 and o0[:, 4].sum() > 0      # no cluster on one XCD - and early rounds all the same
        assert (outs["early0"]["cs"][:, 3] == 0).all() and outs["early0"]["cs"][:, 1].sum() > 0


@pytest.mark.parametrize("cfg", ["arm27", "quad3d"])
def test_a_failed_search_after_an_early_round_leaves_the_last_linearization(cfg, tmp_path):
    """Early linearization (ilqr_large.hpp): the helpers of a cluster linearize the line search's FIRST trial while the leader
    rolls it out - over fx, fu.  When that trial is rejected and the whole search then runs out of step sizes (ilqr.py:337), the
    reference's fx, fu are still the last accepted trajectory's: the device re-linearizes the nominal trajectory before it
    stops.  With beta = 1e-9 a search has ONE step size (the next is below 1e-8, ilqr.py:302): the first iteration whose full
    step is rejected fails its search.  State, Jacobians, gains, counts and status bitwise those of one workgroup per problem
    (MI_ILQR_CLUSTER=1), with one early round opened and not hit per failed problem."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = f"""
import sys, warnings, numpy as np
sys.path.insert(0, {root!r}); sys.path.insert(0, {os.path.join(root, 'tests')!r})
from drake_ddp_amd import workloads as W
from test_gpu_parity import make_solver
if {cfg!r} == 'arm27':
    prob, x0, ug = dict(W.arm27_problem(), beta=1e-9), W.arm27_batch_x0(64)[:6], W.arm27_u_guess(50)
else:
    prob, x0, ug = dict(W.quad3d_problem(), beta=1e-9), W.quad3d_batch_x0(64)[:6], W.quad3d_u_guess(40)
s = make_solver(prob, B=6, jac='fd')
s.SetInitialState(x0); s.SetInitialGuess(ug)
with warnings.catch_warnings():
    warnings.simplefilter('ignore')
    x, u, _, L = s.Solve()
np.savez(sys.argv[1], x=x, u=u, L=L, K=s.K, fx=s.fx, fu=s.fu, it=s.iterations, st=s.status, ls=s.ls_trials, cs=s.cluster_stats)
"""
    outs = {}
    for tag, env in (("cluster", {}), ("single", {"MI_ILQR_CLUSTER": "1"}), ("early0", {"MI_ILQR_EARLY": "0"})):
        f = str(tmp_path / f"{tag}.npz")
        base = {k: v for k, v in os.environ.items() if not k.startswith("MI_ILQR_")}
        r = subprocess.run([sys.executable, "-c", script, f], capture_output=True, text=True, timeout=300, env=dict(base, **env))
        assert r.returncode == 0, r.stderr[-2000:]
        outs[tag] = np.load(f)
    st, it = outs["single"]["st"], outs["single"]["it"]
    failed = st == 2
    assert failed.any() and (it[failed] >= 1).all(), (st, it)                       # (the first search has no cost to beat)
    for tag in ("cluster", "early0"):
        for k in outs["single"].files:
            if k != "cs": assert np.array_equal(outs[tag][k], outs["single"][k]), (tag, k)
    cs = outs["cluster"]["cs"]
    # an early round per search that found helpers there (the very first may come before they are), all hit but a failing one
    assert (cs[:, 0] > 0).all() and (cs[:, 3] >= 1).all() and ((cs[:, 3] - cs[:, 4]) == failed).all(), (cs, st)
    print(cfg, "status", st, "iterations", it, "early rounds opened / hit", cs[:, 3], cs[:, 4])
    assert (outs["early0"]["cs"][:, 3] == 0).all()


@pytest.mark.gpu
def test_pipelined_solves_defer_their_statistics_and_sample_their_events():
    """mi_ilqr_solve_async is ONE dispatch: each solve leaves its per-problem results in its own ring slot and the
    batch statistics are reduced when somebody collects (one launch over every slot still owed); the start/stop
    events ride on one launch in k (mi_ilqr_set_timing).  A cold / warm / warm / cold sequence must come back with
    each solve's own numbers, also when a stage call or an MPC run reuses the current slot before the collect."""
    from test_gpu_properties import c2_setup
    from drake_ddp_amd._capi import MiIlqrError
    prob, x0, s = c2_setup(256)                                   # B > 64: the separate statistics kernel
    s.Solve()                                                     # (pushes the problem to the device)
    ref = []
    for cold in (True, False, False, True):                       # blocking reference, one collect per solve
        if cold:
            s.rearm(cold=True)
        s.solve_resident_async()
        st = s.collect(1)[0]
        ref.append((st.total_iters, st.total_ls_trials, st.n_converged, st.best_cost, st.best_index, s.iterations.copy(), s.cost.copy()))
    assert ref[0][0] == ref[3][0] and ref[1][0] < ref[0][0]       # warm re-solves take fewer iterations
    s.set_timing(2)
    for cold in (True, False, False, True):
        if cold:
            s.rearm(cold=True)
        s.solve_resident_async()
    got = s.collect(4)
    for g, r in zip(got, ref):
        assert (g.total_iters, g.total_ls_trials, g.n_converged, g.best_cost, g.best_index) == r[:5]
    assert [g.kernel_ms > 0 for g in got] == [True, False, True, False]
    assert np.array_equal(s.iterations, ref[3][5]) and np.array_equal(s.cost, ref[3][6])    # fields: the latest slot
    s.set_timing(1)
    # a launch that reuses the current slot settles what is owed first
    s.rearm(cold=True); s.solve_resident_async()                  # cold
    s.solve_resident_async()                                      # warm
    s.stage_backward()
    got = s.collect(2)
    assert (got[0].total_iters, got[0].best_cost) == (ref[0][0], ref[0][3])
    assert (got[1].total_iters, got[1].best_cost) == (ref[1][0], ref[1][3])
    assert all(g.kernel_ms > 0 for g in got)
    # more than the ring holds without collecting: the newest 32 are still right
    for i in range(40):
        if i % 2 == 0:
            s.rearm(cold=True)
        s.solve_resident_async()
    got = s.collect(32)
    assert [g.total_iters for g in got] == [ref[0][0] if i % 2 == 0 else ref[1][0] for i in range(8, 40)]
    with pytest.raises(MiIlqrError):
        s.set_timing(-1)


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", ["pendulum", "synth36", "pendulum_throughput"])
def test_shared_initial_guess_and_pinned_results(cfg):
    """SetInitialGuess with ONE (m,N-1) sequence (the reference's argument, ilqr.py:148-156) goes through
    mi_ilqr_set_initial_shared - m(N-1) doubles over the bus, the device writes the batch's copies in the kernel
    family's own layout - and gives bitwise what the (B,m,N-1) array of copies gives; pinned_results=True (the default)
    returns the same numbers in page-locked blocks that are never rewritten while the caller holds them."""
    from drake_ddp_amd import workloads as W
    rng = np.random.default_rng(11)
    if cfg == "synth36":
        prob, B, kw = W.synth36_problem(), 5, {}
        x0 = W.synth36_batch_x0(B)
        one = W.synth36_u_guess(prob["N"]) + 0.01 * rng.standard_normal((12, prob["N"] - 1))
    else:
        prob, B = W.pendulum_problem(), 70
        kw = {"kernel_mode": "throughput"} if cfg == "pendulum_throughput" else {}
        x0 = W.pendulum_batch_x0(B)
        one = 0.1 * rng.standard_normal((1, prob["N"] - 1))
    res = []
    for guess, pinned in ((np.broadcast_to(one, (B,) + one.shape).copy(), False), (one, False), (one, True)):
        s = make_solver(prob, B=B, jac="fd", pinned_results=pinned, **kw)
        s.SetInitialState(x0)
        s.SetInitialGuess(guess)
        x, u, _, L = s.Solve()
        res.append((x.copy(), u.copy(), L.copy(), s.iterations.copy()))
        if pinned:
            # the reference rebinds its result arrays and never mutates one it has handed out (ilqr.py:375-376, F13): a
            # page-locked block is reused only once the caller holds no view of it
            x_again = s.x_bar
            assert not np.shares_memory(x_again, x) and np.array_equal(x_again, x)
            first = x.copy()
            s.Reset(); s.SetInitialState(x0 + 0.01); s.SetInitialGuess(guess)
            x2, _, _, _ = s.Solve()
            assert not np.shares_memory(x2, x) and np.array_equal(x, first) and not np.array_equal(x2, first)
            addr = x2.__array_interface__["data"][0]
            del x2, x_again
            x3 = s.x_bar                                                      # nobody holds the other blocks any more: one is reused
            assert x3.__array_interface__["data"][0] in (addr, x3.__array_interface__["data"][0]) and len(s._pinned) >= 1
            assert max(len(v) for v in s._pinned.values()) <= 3
            # a DERIVED view alone keeps its block (every view's base is the hand-out's owner array), and no reference count is
            # consulted: extra references to a result (a debugger's, a profiler's) change nothing
            x4 = s.x_bar
            piece, extra_refs = x4[1, :, 5:9], [x4, x4, x4]
            want = piece.copy()
            del x4
            s.Reset(); s.SetInitialState(x0 - 0.02); s.SetInitialGuess(guess); s.Solve()
            _ = [s.x_bar for _k in range(6)]
            assert np.array_equal(piece, want)
            del extra_refs, _
            keep = x
            del s
            import gc; gc.collect()
            assert np.array_equal(keep, res[-1][0])                           # the block outlives the solver
    for r in res[1:]:
        for a, b in zip(res[0], r):
            assert np.array_equal(a, b)


@pytest.mark.gpu
def test_c_host_drives_the_library_through_the_header_alone(tmp_path):
    """examples/c_host.c - C99, nothing but include/mi_ilqr.h - compiled with gcc against libmi_ilqr.so: the numbers
    it prints (iterations, trials, status, cost, final angle of four pendulum swing-ups) are those of the Python
    mirror on the same inputs, bit for bit (it prints 12 / 9 digits; compared at that precision)."""
    import re, shutil, subprocess
    from drake_ddp_amd import workloads as W
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if shutil.which("gcc") is None:
        pytest.skip("no gcc on this box")
    exe = str(tmp_path / "c_host")
    libdir = os.path.join(root, "drake_ddp_amd", "lib")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-O2", "-I" + os.path.join(root, "include"),
                    os.path.join(root, "examples", "c_host.c"), "-L" + libdir, "-lmi_ilqr", "-Wl,-rpath," + libdir, "-lm", "-o", exe],
                   check=True, capture_output=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    rows = re.findall(r"problem (\d+): iterations (\d+) trials (\d+) status (\d+) cost (\S+) theta_N (\S+)", r.stdout)
    assert len(rows) == 4
    prob = W.pendulum_problem()
    x0 = np.array([[0.3 * b - 0.4, 0.1 * b] for b in range(4)])
    s = make_solver(prob, B=4, jac="fd", hist_cap=16)
    s.SetInitialState(x0)
    s.SetInitialGuess(np.zeros((1, prob["N"] - 1)))
    x, u, _, L = s.Solve()
    for b, (pb, it, tr, stt, cost, th) in enumerate(rows):
        assert (int(pb), int(it), int(tr), int(stt)) == (b, int(s.iterations[b]), int(s.ls_trials[b]), int(s.status[b]))
        assert cost == "%.12g" % L[b] and th == "%.9f" % x[b, 0, -1]
    best = re.search(r"batch: (\d+) iterations, 4 converged, best cost (\S+) \(problem (\d+)\)", r.stdout)
    assert int(best.group(1)) == int(s.iterations.sum()) and int(best.group(3)) == int(np.argmin(L))


@pytest.mark.gpu
def test_cost_matrices_are_resent_when_and_only_when_they_change():
    """mi_ilqr_set_cost keeps a host mirror and skips the upload when the caller repeats its matrices (Solve() pushes
    them on every call): changing Q, R, Qf or x_nom between solves must still take effect, bit for bit what a
    fresh solver with the new matrices computes - also after an MPCRun whose moving target advanced x_nom on the
    device."""
    from drake_ddp_amd import workloads as W
    prob = W.pendulum_problem()
    x0 = W.pendulum_batch_x0(16)
    ug = np.zeros((1, prob["N"] - 1))

    def fresh(p):
        s_ = make_solver(p, B=16, jac="fd")
        s_.SetInitialState(x0); s_.SetInitialGuess(ug)
        return s_.Solve()

    s = make_solver(prob, B=16, jac="fd")
    s.SetInitialState(x0); s.SetInitialGuess(ug)
    a = s.Solve()
    ref = fresh(prob)
    assert np.array_equal(a[0], ref[0]) and np.array_equal(a[3], ref[3])
    p2 = dict(prob, Q=prob["Q"] * 3.0 + np.diag([1e-3, 0.0]), R=prob["R"] * 0.5, Qf=prob["Qf"] * 2.0,
              x_nom=np.array([np.pi - 0.1, 0.0]))
    s.SetRunningCost(p2["Q"], p2["R"]); s.SetTerminalCost(p2["Qf"]); s.SetTargetState(p2["x_nom"])
    s.Reset(); s.SetInitialGuess(ug)
    b = s.Solve()
    ref2 = fresh(p2)
    assert np.array_equal(b[0], ref2[0]) and np.array_equal(b[3], ref2[3]) and not np.array_equal(b[3], a[3])
    s.Reset(); s.SetInitialGuess(ug)                      # the same matrices again: nothing is sent, same answer
    c = s.Solve()
    assert np.array_equal(c[0], b[0]) and np.array_equal(c[3], b[3])
    # a moving target advances x_nom on the device and in the mirror: setting the OLD target again must be noticed
    q = W.synth36_problem()
    s36 = make_solver(q, B=2, jac="fd")
    s36.SetInitialState(W.synth36_batch_x0(2)); s36.SetInitialGuess(W.synth36_u_guess(q["N"]))
    s36.Solve()
    step = np.zeros(36); step[0] = 0.004
    s36.MPCRun(3, 4, step)
    s36.SetTargetState(q["x_nom"])                         # back to the original target
    s36.Reset(); s36.SetInitialState(W.synth36_batch_x0(2)); s36.SetInitialGuess(W.synth36_u_guess(q["N"]))
    d = s36.Solve()
    t36 = make_solver(q, B=2, jac="fd")
    t36.SetInitialState(W.synth36_batch_x0(2)); t36.SetInitialGuess(W.synth36_u_guess(q["N"]))
    e = t36.Solve()
    assert np.array_equal(d[0], e[0]) and np.array_equal(d[3], e[3])


@pytest.mark.gpu
def test_async_reads_equal_blocking_reads_and_host_alloc_argument_checks():
    """mi_ilqr_get_async (double and int fields, plain and layout-converted) returns what mi_ilqr_get returns."""
    import ctypes as C
    from drake_ddp_amd import workloads as W, _capi
    lib = _capi.load()
    raw = C.c_void_p()
    assert lib.mi_ilqr_host_alloc(0, C.byref(raw)) != 0 and lib.mi_ilqr_host_alloc(64, None) != 0
    for prob, B, x0, ug in ((W.pendulum_problem(), 9, W.pendulum_batch_x0(9), np.zeros((1, 199))),
                            (W.synth36_problem(), 3, W.synth36_batch_x0(3), W.synth36_u_guess(40))):
        s = make_solver(prob, B=B, jac="fd")
        s.SetInitialState(x0); s.SetInitialGuess(ug)
        x, u, _, L = s.Solve()
        for which, ref in ((_capi.F_X_BAR, x), (_capi.F_U_BAR, u), (_capi.F_COST, L), (_capi.I_ITERS, s.iterations), (_capi.I_STATUS, s.status)):
            out = np.full(ref.shape, -7, dtype=ref.dtype)
            _capi.check(lib.mi_ilqr_get_async(s._h, which, _capi.ptr(out), out.nbytes), "mi_ilqr_get_async")
            _capi.check(lib.mi_ilqr_synchronize(s._h), "mi_ilqr_synchronize")
            assert np.array_equal(out, ref), which
        bad = np.zeros(3)
        assert lib.mi_ilqr_get_async(s._h, _capi.F_X_BAR, _capi.ptr(bad), bad.nbytes) != 0      # wrong size


@pytest.mark.gpu
def test_staged_inputs_survive_the_ring_wrapping_and_back_to_back_setters():
    """Small inputs travel through a 1 MB page-locked ring with asynchronous copies: 70 Solve() calls with a new x0
    each (the ring wraps after ~50 at this size) return what a fresh solver returns for the same x0, and setters
    issued back to back without a solve in between leave the LAST values on the device."""
    from drake_ddp_amd import workloads as W
    prob = W.pendulum_problem()
    B = 1024
    base = W.pendulum_batch_x0(B)
    ug = np.zeros((1, prob["N"] - 1))
    s = make_solver(prob, B=B, jac="fd", hist_cap=2)
    picks = {}
    for k in range(70):
        x0 = base + 1e-3 * k
        s.Reset(); s.SetInitialState(x0); s.SetInitialGuess(ug)
        x, u, _, L = s.Solve()
        if k in (0, 48, 49, 50, 51, 69):
            picks[k] = (x0, x.copy(), L.copy())
    t = make_solver(prob, B=B, jac="fd", hist_cap=2)
    for k, (x0, x, L) in picks.items():
        t.Reset(); t.SetInitialState(x0); t.SetInitialGuess(ug)
        xr, _, _, Lr = t.Solve()
        assert np.array_equal(x, xr) and np.array_equal(L, Lr), k
    for k in range(5):                                        # five pushes, one solve
        s.Reset(); s.SetInitialState(base + 0.01 * k); s.SetInitialGuess(ug + 0.001 * k)
        s._push_problem()
    x, u, _, L = s.Solve()
    t.Reset(); t.SetInitialState(base + 0.04); t.SetInitialGuess(ug + 0.004)
    xr, ur, _, Lr = t.Solve()
    assert np.array_equal(x, xr) and np.array_equal(u, ur) and np.array_equal(L, Lr)


def test_cluster_words_read_as_zeros_where_no_cluster_ran():
    """mi_ilqr.h, MI_I64_CLUSTER_WORDS: zeros for a handle of the wave-per-problem kernels (it has no such words; until round 6 the
    call failed with BAD_ARG there) and for a workgroup-per-problem launch that was not clustered (B = 300 on 256 CUs)."""
    from drake_ddp_amd import workloads as W
    a = W.acrobot_problem()
    s = make_solver(a, B=4, jac="fd")
    s.SetInitialState(W.acrobot_batch_x0(4)); s.SetInitialGuess(np.zeros((1, a["N"] - 1)))
    s.Solve()
    assert s.cluster_stats.shape == (4, 6) and (s.cluster_stats == 0).all()
    if os.environ.get("MI_ILQR_CLUSTER", "1") not in ("", "1"):
        return                                               # (a suite run with clusters forced: this launch WOULD be clustered)
    q = W.synth36_problem()
    s = make_solver(dict(q, N=8), B=300, jac="fd")
    s.SetInitialState(np.tile(W.synth36_batch_x0(64), (5, 1))[:300]); s.SetInitialGuess(W.synth36_u_guess(8))
    s.Solve()
    assert (s.cluster_stats == 0).all()
