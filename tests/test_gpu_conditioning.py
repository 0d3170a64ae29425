"""What the workgroup-per-problem backward passes (ilqr_large.hpp: large_backward n = 36 / 37, mid_backward n <= 32) do when
Quu is ILL-CONDITIONED or NOT POSITIVE DEFINITE.  The reference inverts Quu with LU + partial pivoting and no regularization
(np.linalg.inv, ilqr.py:655); the device eliminates without pivoting (Gauss-Jordan, one row per lane) - exact for a positive
definite matrix, and every pivot is checked: a Quu that is not positive definite stops its problem with MI_STATUS_NOT_PD
instead of producing gains silently.  Yardstick for accuracy: the reference's recursion in extended precision
(tests/common.py: backward_extended) - the fp64 NumPy oracle's own distance from it is what the problem's conditioning
allows, the device must stay within 20 x that (or 1e-11, SURVEY 8(c)'s stage-level figure)."""
import os
import sys

import numpy as np
import pytest

from common import backward_errors, load_golden, make_oracle
from test_gpu_parity import make_solver

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "examples", "plugins"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

STAGES = ["synth36_stage", "quad3d_stage", "quad_stage", "arm27_stage"]


def _backward_on_golden_inputs(g, prob, Q=None, R=None, Qf=None):
    """Device and NumPy backward pass on the fixture's own trajectory and Jacobians (identical inputs), optionally with other
    cost matrices.  Returns (solver, oracle)."""
    prob = dict(prob, Q=prob["Q"] if Q is None else Q, R=prob["R"] if R is None else R, Qf=prob["Qf"] if Qf is None else Qf)
    s = make_solver(prob, jac="ad")
    s.SetInitialState(g["x0"][None])
    s.SetInitialGuess(g["roll_u"])
    s.set_state(x_bar=g["roll_x"][None], u_bar=g["roll_u"][None], fx=g["fx"][None], fu=g["fu"][None])
    s.stage_backward()
    o = make_oracle(prob)
    o.set_problem(g["x0"], prob["x_nom"], prob["Q"], prob["R"], prob["Qf"], g["roll_u"])
    o.x_bar, o.fx, o.fu = g["roll_x"].copy(), g["fx"].copy(), g["fu"].copy()
    o.backward()
    return s, o


@pytest.mark.parametrize("name", STAGES)
@pytest.mark.parametrize("rscale", [1.0, 1e-3, 1e-6, 1e-9])
def test_backward_pass_with_vanishing_control_cost(name, rscale):
    """R = rscale x the fixture's R (down to 1e-9 of it: R = 1e-11 * dt * I on the quadrupeds) on the reference-recorded
    trajectories and Jacobians of the 36-state chain, the two quadrupeds (contact) and the arm + ball.  Gains against the
    extended-precision recursion, judged by the fp64 reference's own distance from it."""
    g, prob = load_golden(name)
    s, o = _backward_on_golden_inputs(g, prob, R=rscale * prob["R"])
    e_dev, e_ref, cond = backward_errors((s.K[0], s.kappa[0], s.dV_coeff[0]), o)
    st = int(s.status[0])
    print(f"{name} R x {rscale:g}: device {e_dev:.2e}, NumPy fp64 {e_ref:.2e} from the extended-precision pass; max cond(Quu) {cond:.1e}; "
          f"max|K| {np.abs(o.K).max():.1e}; status {st}")
    # Observed (MI355X, round 4): with the fixtures' own R every pass is at round-off level (6e-15 .. 6e-10, the device
    # closer to the extended-precision result than NumPy in three of four).  With R -> 0 the RECURSION loses its digits, not
    # the elimination (cond(Quu) stays below 1e6): Vxx' = Qxx - Qux^T Quu^{-1} Qux cancels catastrophically, and the fp64
    # NumPy pass itself ends 4e-2 .. 1.1 away from the extended-precision one - i.e. the reference's gains are noise there.
    # The device stays within the yardstick wherever it reports success; where round-off has made a Quu lose positive
    # definiteness it says so (arm27, R x 1e-9) - allowed only where the fp64 reference has itself lost every digit.
    from drake_ddp_amd import _capi
    if st == _capi.STATUS_NOT_PD:
        assert e_ref > 1e-2, (e_ref, cond)
    else:
        assert st == 0 and np.isfinite(s.K).all() and e_dev < max(1e-11, 20 * e_ref), (e_dev, e_ref, cond)
    if rscale == 1.0:
        assert st == 0 and e_dev < 1e-9


def _indefinite(rng, A, neg):
    """A symmetric matrix with A's eigenvectors scrambled and `neg` of its eigenvalues made negative."""
    n = A.shape[0]
    w = np.sort(np.abs(np.linalg.eigvalsh(A)) + 1e-3 * np.abs(A).max())[::-1].copy()
    w[:neg] *= -1.0
    Qm, _ = np.linalg.qr(rng.standard_normal((n, n)))
    M_ = (Qm * w) @ Qm.T
    return 0.5 * (M_ + M_.T)


@pytest.mark.parametrize("name", STAGES)
def test_indefinite_terminal_cost_matches_or_is_reported(name):
    """A dense symmetric Qf with ONE NEGATIVE eigenvalue makes Vxx indefinite from the first step on.  The reference accepts
    any matrices (ilqr.py:136-146) and inverts whatever Quu comes out.  Here: while every Quu = 2R + fu^T Vxx fu of the pass
    stays positive definite (large R) the gains must match the extended-precision recursion like any others; once one does
    not (small R: the indefinite direction reaches the inputs) the device says so - status MI_STATUS_NOT_PD - and does not
    hand out gains as if nothing happened.  Both regimes are established in the test by the eigenvalues of the oracle's Quu."""
    from drake_ddp_amd import _capi
    g, prob = load_golden(name)
    rng = np.random.default_rng(7)
    Qf = _indefinite(rng, prob["Qf"], 1)
    n, m, N = prob["Q"].shape[0], prob["R"].shape[0], prob["N"]

    def min_quu_eig(R):
        o = make_oracle(dict(prob, R=R, Qf=Qf))
        o.set_problem(g["x0"], prob["x_nom"], prob["Q"], R, Qf, g["roll_u"])
        o.x_bar, o.fx, o.fu = g["roll_x"].copy(), g["fx"].copy(), g["fu"].copy()
        Vxx, lo = 2 * Qf, np.inf
        for t in range(N - 2, -1, -1):
            fx, fu = o.fx[:, :, t], o.fu[:, :, t]
            Quu = 2 * R + fu.T @ Vxx @ fu
            Qux = fu.T @ Vxx @ fx
            lo = min(lo, float(np.linalg.eigvalsh(0.5 * (Quu + Quu.T)).min()) / float(np.abs(Quu).max()))
            Vxx = 2 * prob["Q"] + fx.T @ Vxx @ fx - Qux.T @ np.linalg.inv(Quu) @ Qux
        return lo

    seen = set()
    for rs in (1e12, 1e9, 1e6, 1e4, 1e2, 1.0, 1e-2, 1e-4):
        R = rs * prob["R"]
        lo = min_quu_eig(R)
        if abs(lo) < 1e-6:                    # (too close to singular to call either way)
            continue
        s, o = _backward_on_golden_inputs(g, prob, R=R, Qf=Qf)
        st = int(s.status[0])
        if lo > 0:
            e_dev, e_ref, cond = backward_errors((s.K[0], s.kappa[0], s.dV_coeff[0]), o)
            print(f"{name} indefinite Qf, R x {rs:g}: every Quu positive definite (min eig / max entry {lo:.1e}); device {e_dev:.2e}, NumPy {e_ref:.2e}; cond {cond:.1e}")
            assert st == 0 and e_dev < max(1e-11, 20 * e_ref)
            seen.add("pd")
        else:
            print(f"{name} indefinite Qf, R x {rs:g}: a Quu with a negative eigenvalue ({lo:.1e} of its largest entry): status {st}")
            assert st == _capi.STATUS_NOT_PD
            seen.add("not_pd")
    assert seen == {"pd", "not_pd"}, seen


def test_solve_stops_with_not_pd_instead_of_silent_garbage():
    """End to end: an indefinite terminal cost on the arm + ball problem.  The solve stops at the first backward pass that
    meets a Quu which is not positive definite: per-problem status MI_STATUS_NOT_PD, counted in stats.n_not_pd, raised as
    RuntimeError by both classes - while the same problem with its regular cost still solves (the check costs nothing there).
    Asymmetric matrices stay refused at mi_ilqr_set_cost (E_UNSUPPORTED) on these kernels."""
    from drake_ddp_amd import _capi, workloads as W
    prob = W.arm27_problem()
    rng = np.random.default_rng(3)
    Qf = _indefinite(rng, prob["Qf"], 3)
    bad = dict(prob, Qf=Qf, R=1e-3 * prob["R"])
    s = make_solver(bad, B=3, jac="fd")
    s.SetInitialState(W.arm27_batch_x0(3)); s.SetInitialGuess(W.arm27_u_guess(prob["N"]))
    with pytest.raises(RuntimeError, match="not positive definite"):
        s.Solve()
    assert (s.status == _capi.STATUS_NOT_PD).all() and s.stats.n_not_pd == 3 and s.stats.n_converged == 0
    one = make_solver(bad, jac="fd", single=True)
    one.SetInitialState(W.arm27_start()); one.SetInitialGuess(W.arm27_u_guess(prob["N"]))
    with pytest.raises(RuntimeError, match="not positive definite"):
        one.Solve()
    ok = make_solver(prob, B=3, jac="fd")
    ok.SetInitialState(W.arm27_batch_x0(3)); ok.SetInitialGuess(W.arm27_u_guess(prob["N"]))
    ok.Solve()
    assert (ok.status == 0).all() and ok.stats.n_not_pd == 0
    asym = prob["Q"].copy()
    asym[0, 1] += 1e-3
    t = make_solver(dict(prob, Q=asym), B=1, jac="fd")
    t.SetInitialState(W.arm27_start()[None]); t.SetInitialGuess(W.arm27_u_guess(prob["N"]))
    with pytest.raises(Exception, match="not supported"):
        t.Solve()


def test_riccati_error_growth_with_the_horizon_and_on_indefinite():
    """What a LONG horizon does to the backward pass of a stiff contact model (the planar quadruped, dt = 1.5e-3): cond(Quu) stays
    ~2e3, but the recursion amplifies round-off ~10 x every dozen steps - against the extended-precision pass the fp64 reference
    is at 1e-8 (N = 40), 1e-5 (80), 0.2 (110) and the device at 1e-10, 1e-6, 1e-3: the device keeps one to two more digits for
    as long as anybody has any.  Past that (N = 148) round-off makes a Quu indefinite: the reference inverts it all the same
    (ilqr.py:655) and carries on; the device stops the problem with MI_STATUS_NOT_PD by default and carries on like the reference
    with on_indefinite = "continue" (mi_ilqr_desc.on_indefinite) - never an error, never a hang."""
    from drake_ddp_amd import _capi, workloads as W
    got = {}
    for N in (40, 80, 110):
        p = dict(W.planar_quad_problem(), dt=1.5e-3, N=N)
        x0, ug = W.planar_quad_batch_x0(1), W.planar_quad_u_guess(N)
        s = make_solver(p, B=1, jac="fd")
        s.SetInitialState(x0); s.SetInitialGuess(ug)
        s.stage_forward(np.inf); s.stage_linearize()
        xb, ub, fx, fu = s.x_bar[0], s.u_bar[0], s.fx[0], s.fu[0]
        s.stage_backward()
        o = make_oracle(p)
        o.set_problem(x0[0], p["x_nom"], p["Q"], p["R"], p["Qf"], ub)
        o.x_bar, o.u_bar, o.fx, o.fu = xb.copy(), ub.copy(), fx.copy(), fu.copy()
        o.backward()
        e_dev, e_ref, cond = backward_errors((s.K[0], s.kappa[0], s.dV_coeff[0]), o)
        got[N] = (e_dev, e_ref)
        print(f"N = {N}: device {e_dev:.1e}, NumPy fp64 {e_ref:.1e} from the extended-precision pass; max cond(Quu) {cond:.1e}")
        assert int(s.status[0]) == 0 and e_dev < max(1e-9, 2 * e_ref)
    assert got[110][1] > 1e3 * got[40][1]                               # (the growth is the recursion's, not the elimination's)
    p = dict(W.planar_quad_problem(), dt=1.5e-3, N=148)
    x0, ug = W.planar_quad_batch_x0(4), W.planar_quad_u_guess(148)
    stop = make_solver(p, B=4, jac="fd")
    stop.SetInitialState(x0); stop.SetInitialGuess(ug)
    with pytest.raises(RuntimeError, match="not positive definite"):
        stop.Solve()
    assert (stop.status == _capi.STATUS_NOT_PD).all()
    go = make_solver(p, B=4, jac="fd", on_indefinite="continue")
    go.SetInitialState(x0); go.SetInitialGuess(ug)
    go.Solve()
    print(f"N = 148: default -> status {stop.status.tolist()} after {stop.iterations.tolist()} iterations; on_indefinite = continue -> status {go.status.tolist()}, "
          f"{go.iterations.tolist()} iterations, costs {np.round(go.cost, 3).tolist()} (first rollout: {np.round(stop.history[:, 0, 0], 3).tolist()})")
    assert (go.status != _capi.STATUS_NOT_PD).all() and np.isfinite(go.cost).all() and (go.cost <= stop.history[:, 0, 0]).all()
