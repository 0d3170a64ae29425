"""The MID-SIZE workgroup-per-problem family (ilqr_large.hpp: mid_backward): 4 < n <= 32 with ANY m <= 16 - the shapes the
reference class accepts (ilqr.py:57-58 takes whatever the system reports) that neither the wave-per-problem kernels
(m <= 2) nor the n = 36 / 37 kernels cover: a quadrotor's (12, 4), a 7-joint arm's (14, 7), kinova_gen3.py's arm + free body
(27, 7), the family's corners (32, 16), (16, 1), odd state dimensions (7, 3), (9, 4).  Plugin models of those shapes
(examples/plugins/models.py: chainx) against the NumPy oracle driven by the same update in Python (tests/plugin_steps.py)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "examples", "plugins"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

SHAPES = [(6, 4, 0), (7, 7, 0), (10, 7, 7), (16, 16, 0), (4, 4, 1), (3, 3, 1), (8, 1, 0)]    # (nq, m, ne): n = 2 nq + ne
IDS = ["n%d_m%d" % (2 * nq + ne, m) for nq, m, ne in SHAPES]


def test_mid_family_registers_the_reviewed_shapes():
    """mi_ilqr_register_model accepts (n, m) in {(12, 4), (14, 7), (27, 7)} and the family's corners on family 1; a model with
    more than 16 inputs, or 32 < n <= 40 outside the large kernels' wave roles, is still refused.  No device needed."""
    import ctypes as C
    import models as PM
    from drake_ddp_amd import _capi, plugin
    assert PM.CHAINX_SHAPES == SHAPES and PM.PADDED_SHAPES == PADDED and PM.LARGE_SHAPES == LARGE
    make = PM.build_all()
    for nq, m, ne in SHAPES:
        s = make["chainx_%d_%d_%d" % (nq, m, ne)](0.01)
        assert s.model_id >= 100 and (s.n, s.m) == (2 * nq + ne, m)
    lib = _capi.load()
    good = plugin._Plugin()
    plug = C.CDLL(plugin.plugin_path("vdp", plugin.source("vdp", 2, 1, PM.VDP_BODY, PM.VDP_DEFAULTS)))
    plug.mi_plugin_describe.argtypes = [C.POINTER(plugin._Plugin)]
    plug.mi_plugin_describe(C.byref(good))
    mid = C.c_int32()
    for n, m, ok in ((12, 17, False), (36, 10, False), (36, 20, False), (41, 4, False)):
        rec = plugin._Plugin(abi_version=good.abi_version, kernel_args_bytes=good.kernel_args_bytes, handle_bytes=good.handle_bytes, n=n, m=m, n_params=0, family=1, launch=1, lds_bytes=1)
        rc = lib.mi_ilqr_register_model(C.byref(rec), C.byref(mid))
        assert rc in (_capi.E_UNSUPPORTED, _capi.E_BAD_SHAPE), (n, m, rc)
    assert plugin.resolve_family(27, 7, "auto") == "large" and plugin.resolve_family(4, 1, "auto") == "small"


def _random_spd(rng, k, lo, hi):
    A = rng.standard_normal((k, k))
    Qm, _ = np.linalg.qr(A)
    return (Qm * 10.0 ** rng.uniform(lo, hi, k)) @ Qm.T


@pytest.mark.gpu
@pytest.mark.parametrize("shape", SHAPES, ids=IDS)
def test_mid_backward_pass_on_random_inputs(shape):
    """Stage level with identical inputs: random trajectories, random Jacobians (identity + noise), DENSE random symmetric
    positive definite Q, R, Qf over four decades; five horizons including the shortest the library accepts (every start-up
    branch of F's pipeline).  K, kappa, dV against the reference's recursion in EXTENDED precision (tests/common.py):
    1e-11 or 20 x the fp64 NumPy oracle's own distance from it, whichever is larger - observed on MI355X:
    the device is closer to the extended-precision result than NumPy's fp64 pass in six of the seven shapes."""
    import models as PM
    import plugin_steps as PS
    from drake_ddp_amd.ilqr import BatchedIterativeLQR
    from oracle import models_np as M
    from oracle.ilqr_np import OracleILQR
    from common import backward_errors
    nq, m, ne = shape
    n = 2 * nq + ne
    sys_ = PM.build_chainx(nq, m, ne)(0.02)
    model = M.Model.custom(n, m, PS.chainx_step(nq, m, ne), sys_.params, 0.02)
    rng = np.random.default_rng(100 * n + m)
    worst = worst_ref = worst_cond = 0.0
    for N in (4, 5, 6, 7, 33):
        B = 3
        Q, R, Qf = _random_spd(rng, n, -3, 0), _random_spd(rng, m, -2.5, 0), _random_spd(rng, n, -1, 1)
        Q, R, Qf = 0.5 * (Q + Q.T), 0.5 * (R + R.T), 0.5 * (Qf + Qf.T)
        x_nom = rng.standard_normal(n)
        xb = rng.standard_normal((B, n, N))
        ub = rng.standard_normal((B, m, N - 1))
        fx = np.eye(n)[None, :, :, None] + (0.5 / np.sqrt(n)) * rng.standard_normal((B, n, n, N - 1))
        fu = 0.3 * rng.standard_normal((B, n, m, N - 1))
        s = BatchedIterativeLQR(sys_, N, B, jacobian_mode="ad")
        s.SetTargetState(x_nom); s.SetRunningCost(Q, R); s.SetTerminalCost(Qf)
        s.SetInitialState(xb[:, :, 0]); s.SetInitialGuess(ub)
        s.set_state(x_bar=xb, u_bar=ub, fx=fx, fu=fu)
        s.stage_backward()
        K, kap, dV = s.K, s.kappa, s.dV_coeff
        for b in range(B):
            o = OracleILQR(model, N)
            o.set_problem(xb[b, :, 0], x_nom, Q, R, Qf, ub[b])
            o.x_bar, o.fx, o.fu = xb[b], fx[b], fu[b]
            o.backward()
            # judged against the extended-precision pass: 1e-11 or, where the problem itself leaves fewer
            # digits (cond(Quu), the dynamic range of Vxx over the horizon), 20 x the fp64 NumPy oracle's own distance
            e_dev, e_ref, cond = backward_errors((K[b], kap[b], dV[b]), o)
            worst, worst_ref, worst_cond = max(worst, e_dev), max(worst_ref, e_ref), max(worst_cond, cond)
            assert e_dev < max(1e-11, 20 * e_ref), (shape, N, b, e_dev, e_ref, cond)
    print(f"mid backward {IDS[SHAPES.index(shape)]}: worst relative error vs extended precision {worst:.2e} (NumPy fp64 oracle: {worst_ref:.2e}; max cond(Quu) {worst_cond:.1e})")


PADDED = [(18, 7, 0), (16, 3, 5)]       # n = 36, m = 7 and n = 37, m = 3: the n > 32 kernels with padded controls
# n > 32 with m = 4, 8, 16 (examples/plugins/models.py: LARGE_SHAPES).  Until round 4 only m = 12 had ever run on these kernels,
# and (36, 4), (36, 8), (40, 4) - pad columns behind u inside the shared tile - and every m = 16 (row stride of Quu^{-1}'s
# exchange buffer) returned WRONG gains without any error; found by the sweep of all 32 (n, m) in 33..40 x {4, 8, 12, 16}.
LARGE = [(18, 4, 0), (18, 8, 0), (20, 4, 0), (20, 8, 0), (17, 4, 1), (19, 8, 0), (19, 12, 1), (16, 16, 1), (18, 16, 0), (20, 16, 0)]


@pytest.mark.gpu
@pytest.mark.parametrize("shape", SHAPES + PADDED + LARGE, ids=IDS + ["n36_m7_padded", "n37_m3_padded"] + ["n%d_m%d" % (2 * a + c, b) for a, b, c in LARGE])
def test_mid_family_solves_like_the_oracle(shape):
    """End to end on a plugin model of the shape: forward-mode duals after two iterations (round-off level: x_bar 1e-11,
    K / kappa 1e-10) and at convergence (iterations, step sizes and trial counts exact, cost 1e-10); central differences
    at convergence (counts exact, cost 1e-8, trajectories 1e-6: SURVEY 8(c)).
    The two PADDED shapes close the last hole in n <= 40: above 32 states the matrix-core backward pass wants m % 4 == 0, so
    the plugin's device model carries one padding control that its step never reads (8 for 7, 4 for 3) and the class pads R
    with a unit block and the guess with zeros - the padding's gains, feed-forward terms and inputs are exact zeros and what
    the caller sees has m controls: compared here with the NumPy oracle of the UNPADDED problem."""
    import models as PM
    import plugin_steps as PS
    from drake_ddp_amd.ilqr import BatchedIterativeLQR
    from oracle import models_np as M
    from oracle.ilqr_np import OracleILQR
    nq, m, ne = shape
    n = 2 * nq + ne
    dt, N, B = 0.02, 24, 3
    sys_ = PM.build_chainx(nq, m, ne)(dt)
    assert (sys_.n, sys_.m) == (n, m) and sys_.m_dev == (m if shape not in PADDED else 4 * ((m + 3) // 4))
    rng = np.random.default_rng(n * 17 + m)
    x_nom = np.zeros(n)
    x0 = 0.4 * rng.standard_normal((B, n))
    ug = 0.2 * rng.standard_normal((B, m, N - 1))
    Q = dt * np.diag(10.0 ** rng.uniform(-1, 0.5, n))
    R = dt * 0.05 * np.eye(m)
    Qf = np.diag(10.0 ** rng.uniform(0, 1, n))
    model = M.Model.custom(n, m, PS.chainx_step(nq, m, ne), sys_.params, dt)
    for jac, cap in (("ad", 2), ("ad", 100000), ("fd", 100000)):
        s = BatchedIterativeLQR(sys_, N, B, delta=1e-3, beta=0.7, gamma=0.0, jacobian_mode=jac, **({"max_iters": cap} if cap == 2 else {}))
        s.SetTargetState(x_nom); s.SetRunningCost(Q, R); s.SetTerminalCost(Qf)
        s.SetInitialState(x0); s.SetInitialGuess(ug)
        s.Solve()                                                # (MI_STATUS_MAX_ITERS raises nothing: cap = 2 included)
        for b in range(B):
            o = OracleILQR(model, N, 1e-3, 0.7, 0.0, jacobian=jac, fd_step=1e-5, max_iters=cap)
            o.set_problem(x0[b], x_nom, Q, R, Qf, ug[b])
            xo, uo, Lo, hist = o.solve()
            hist = np.array(hist)
            assert s.iterations[b] == len(hist) and np.array_equal(s.history[b][:len(hist), 1:3], hist[:, 1:3]), (jac, cap, b)
            sc = lambda a_: max(1.0, float(np.max(np.abs(a_))))
            assert s.u_bar.shape == (B, m, N - 1) and s.K.shape == (B, m, n, N - 1) and s.fu.shape == (B, n, m, N - 1)
            if cap == 2:
                assert np.max(np.abs(s.x_bar[b] - xo)) < 1e-11 * sc(xo) and np.max(np.abs(s.K[b] - o.K)) < 1e-10 * sc(o.K)
                assert np.max(np.abs(s.kappa[b] - o.kappa)) < 1e-10 * sc(o.kappa)
            elif jac == "ad":
                assert s.status[b] == 0 and abs(s.cost[b] - Lo) < 1e-10 * abs(Lo)
                assert np.max(np.abs(s.x_bar[b] - xo)) < 1e-8 * sc(xo) and np.max(np.abs(s.K[b] - o.K)) < 1e-7 * sc(o.K)
            else:
                assert s.status[b] == 0 and abs(s.cost[b] - Lo) < 1e-8 * abs(Lo)
                assert np.max(np.abs(s.x_bar[b] - xo)) < 1e-6 * sc(xo) and np.max(np.abs(s.u_bar[b] - uo)) < 1e-6 * sc(uo)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(6, 4, 0), (10, 7, 7), (18, 7, 0)], ids=["n12_m4", "n27_m7", "n36_m7_padded"])
def test_mid_family_mpc_loop_and_keypoints(shape):
    """The receding-horizon loop on the device (mi_ilqr_mpc_run: stale-gain first rollouts) and the key-point
    methods (shared code of the workgroup-per-problem kernels) on mid-size shapes: per-re-solve iteration counts exact,
    costs 1e-8; setInterval(3) key-point lists exact."""
    import models as PM
    import plugin_steps as PS
    from drake_ddp_amd import utils_derivs_interpolation as U
    from drake_ddp_amd.ilqr import BatchedIterativeLQR
    from drake_ddp_amd.workloads import mpc_shift
    from oracle import models_np as M
    from oracle.ilqr_np import OracleILQR, KeypointCfg
    nq, m, ne = shape
    n = 2 * nq + ne
    dt, N, B = 0.02, 30, 4
    sys_ = PM.build_chainx(nq, m, ne)(dt)
    rng = np.random.default_rng(n + 3 * m)
    x_nom = np.zeros(n)
    x0 = 0.3 * rng.standard_normal((B, n))
    ug = 0.1 * rng.standard_normal((m, N - 1))
    Q, R, Qf = dt * np.eye(n), dt * 0.05 * np.eye(m), 5.0 * np.eye(n)
    model = M.Model.custom(n, m, PS.chainx_step(nq, m, ne), sys_.params, dt)
    s = BatchedIterativeLQR(sys_, N, B, delta=1e-3, beta=0.6, gamma=0.0, jacobian_mode="fd")
    s.SetTargetState(x_nom); s.SetRunningCost(Q, R); s.SetTerminalCost(Qf)
    s.SetInitialState(x0); s.SetInitialGuess(ug)
    s.Solve()
    it0 = s.iterations.copy()
    s.MPCRun(4, 3)
    log = s.mpc_log
    for b in range(B):
        o = OracleILQR(model, N, 1e-3, 0.6, 0.0, jacobian="fd", fd_step=1e-5)
        o.set_problem(x0[b], x_nom, Q, R, Qf, ug)
        xo, uo, Lo, hist = o.solve()
        assert len(hist) == it0[b], b
        for r in range(4):
            x0r, ugr = mpc_shift(xo, uo, 3)
            o.set_problem(x0r, x_nom, Q, R, Qf, ugr)
            xo, uo, Lo, hist = o.solve()
            assert log[b, r, -1] == len(hist) and abs(log[b, r, -2] - Lo) < 1e-8 * abs(Lo), (b, r)
            assert np.max(np.abs(log[b, r, :n] - x0r)) < 1e-6
    kp = U.derivs_interpolation("setInterval", 3, 0, 0, 0)
    s = BatchedIterativeLQR(sys_, N, B, delta=1e-3, beta=0.6, gamma=0.0, jacobian_mode="ad", derivs_keypoint_method=kp)
    s.SetTargetState(x_nom); s.SetRunningCost(Q, R); s.SetTerminalCost(Qf)
    s.SetInitialState(x0); s.SetInitialGuess(ug)
    s.Solve()
    for b in range(B):
        o = OracleILQR(model, N, 1e-3, 0.6, 0.0, keypoint=KeypointCfg("setInterval", 3, 0, 0.0, 0.0), jacobian="ad")
        o.set_problem(x0[b], x_nom, Q, R, Qf, ug)
        xo, uo, Lo, hist = o.solve()
        nk = int(s.keypoint_count[b])
        assert len(hist) == s.iterations[b] and list(s.keypoint_list[b][:nk]) == list(o.keypoints), b
        assert abs(s.cost[b] - Lo) < 1e-9 * abs(Lo)


@pytest.mark.gpu
def test_mid_family_batch_position_and_size_invariance():
    """Problems of a batch are independent (the sharding argument): a problem solved alone, in a batch of 5 and in a
    batch of 70 at another position - and with ragged horizons around the F pipeline's start-up branches - gives bitwise the
    same trajectory, gains and cost on the mid-size kernels (arm + ball, central differences; ragged N on a (12, 4) plugin)."""
    import models as PM
    from drake_ddp_amd import workloads as W
    from drake_ddp_amd.ilqr import BatchedIterativeLQR
    from drake_ddp_amd.models import ArmAndBall
    p = W.arm27_problem()
    x0 = W.arm27_batch_x0(70)

    def solve(xs):
        s = BatchedIterativeLQR(ArmAndBall(p["dt"]), p["N"], len(xs), delta=p["delta"], beta=p["beta"], gamma=p["gamma"], max_iters=6)
        s.SetTargetState(p["x_nom"]); s.SetRunningCost(p["Q"], p["R"]); s.SetTerminalCost(p["Qf"])
        s.SetInitialState(xs); s.SetInitialGuess(W.arm27_u_guess(p["N"]))
        try:
            s.Solve()
        except RuntimeError:
            pass
        return s.x_bar.copy(), s.K.copy(), s.cost.copy(), s.iterations.copy()
    whole = solve(x0)
    for lo, hi in ((3, 4), (10, 15), (69, 70)):
        part = solve(x0[lo:hi])
        for a, b in zip(whole, part):
            assert np.array_equal(a[lo:hi], b), (lo, hi)
    assert (whole[3] == 6).all()
    # ragged horizons on a plugin shape with one x-wave: N = 4 (the shortest the library accepts) .. 9
    sys_ = PM.build_chainx(6, 4, 0)(0.02)
    rng = np.random.default_rng(3)
    xs = 0.3 * rng.standard_normal((4, 12))
    for N in (4, 5, 6, 7, 8, 9):
        outs = []
        for B in (1, 4):
            s = BatchedIterativeLQR(sys_, N, B, delta=1e-4, beta=0.6)
            s.SetTargetState(np.zeros(12)); s.SetRunningCost(0.02 * np.eye(12), 0.001 * np.eye(4)); s.SetTerminalCost(3.0 * np.eye(12))
            s.SetInitialState(xs[:B]); s.SetInitialGuess(np.zeros((4, N - 1)))
            s.Solve()
            outs.append((s.x_bar.copy(), s.K.copy(), s.cost.copy()))
        for a, b in zip(*outs):
            assert np.array_equal(a[:1], b[:1]), N


@pytest.mark.gpu
def test_four_candidate_line_search_is_the_sequential_one(tmp_path):
    """mid_rollout4 walks four line-search candidates (eps, eps beta, eps beta^2, eps beta^3) through one step loop and the
    search takes the first that passes (ilqr.py:330) - what the sequential search returns.  Against the same kernels with
    the candidates one after the other (MI_ILQR_SPEC=0), and with four candidates from the first finite cost on (=2), on 96 arm +
    ball problems (2.4 trials per iteration) and a plugin chain that backtracks with beta = 0.5: iterations, trials per
    problem and the eps / trial count of EVERY iteration are identical; costs agree to round-off (the only arithmetic
    difference: after a four-candidate pass the backward pass forms lx, lu from x_bar, u_bar instead of taking the rows
    the rollout left - 2 Q x - 2 Q x_nom against 2 (Q (x - x_nom)))."""
    import subprocess
    script = f"""
import sys, numpy as np
sys.path.insert(0, {ROOT!r}); sys.path.insert(0, {os.path.join(ROOT, 'tests')!r}); sys.path.insert(0, {os.path.join(ROOT, 'examples', 'plugins')!r})
from drake_ddp_amd import workloads as W
from drake_ddp_amd.ilqr import BatchedIterativeLQR
from test_gpu_parity import make_solver
import models as PM
p = W.arm27_problem()
B = 96
s = make_solver(p, B=B, jac='fd', hist_cap=64)
s.SetInitialState(W.arm27_batch_x0(B)); s.SetInitialGuess(W.arm27_u_guess(p['N']))
x, u, _, L = s.Solve()
out = dict(L=L, it=s.iterations, ls=s.ls_trials, h=s.history[:, :, 1:3], x=x, st=s.status)
n, m, N, dt = 12, 4, 40, 0.03
sys_ = PM.build_chainx(6, 4, 0)(dt)
c = BatchedIterativeLQR(sys_, N, 64, delta=1e-4, beta=0.5, jacobian_mode='ad', hist_cap=64)
c.SetTargetState(np.concatenate([np.full(6, 2.5), np.zeros(6)])); c.SetRunningCost(dt * np.eye(n), dt * 1e-3 * np.eye(m)); c.SetTerminalCost(80.0 * np.eye(n))
c.SetInitialState(np.random.default_rng(5).uniform(-1.5, 1.5, (64, n))); c.SetInitialGuess(np.zeros((m, N - 1)))
try:
    xc, uc, _, Lc = c.Solve()
except RuntimeError:
    xc, Lc = c.x_bar, c.cost
out.update(cL=Lc, cit=c.iterations, cls=c.ls_trials, ch=c.history[:, :, 1:3], cst=c.status)
np.savez(sys.argv[1], **out)
"""
    outs = {}
    for pol in ("1", "0", "2"):
        f = str(tmp_path / f"spec{pol}.npz")
        r = subprocess.run([sys.executable, "-c", script, f], capture_output=True, text=True, timeout=600, env=dict(os.environ, MI_ILQR_SPEC=pol))
        assert r.returncode == 0, r.stderr[-2000:]
        outs[pol] = np.load(f)
    ref = outs["0"]
    print(f"arm + ball: {int(ref['ls'].sum())} trials in {int(ref['it'].sum())} iterations; chain: {int(ref['cls'].sum())} in {int(ref['cit'].sum())}, "
          f"status {np.unique(ref['cst']).tolist()}")
    assert ref["ls"].sum() > 1.5 * ref["it"].sum() and ref["cls"].sum() > 1.2 * ref["cit"].sum()      # (both cases do backtrack)
    for pol in ("1", "2"):
        o = outs[pol]
        for k in ("it", "ls", "h", "st", "cit", "cls", "ch", "cst"):
            assert np.array_equal(o[k], ref[k]), (pol, k)
        ok = ref["cst"] == 0
        dev = max(np.max(np.abs(o["L"] - ref["L"]) / np.abs(ref["L"])), np.max(np.abs(o["cL"][ok] - ref["cL"][ok]) / np.abs(ref["cL"][ok])))
        print(f"  MI_ILQR_SPEC={pol}: every decision identical; largest relative cost difference {dev:.1e}")
        assert dev < 1e-9


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["chainx_12_4", "chainx_27_7", "arm27_n16", "arm27_n24",
                                   "synth36_n16", "quad3d_n12", "chainx_40_16", "chainx_36_7", "chainx_33_16", "chainx_40_8"])
def test_asymmetric_costs_follow_the_reference_on_the_mid_size_kernels(which):
    """The reference accepts any Q, R, Qf and never symmetrizes (ilqr.py:120-146,161-186): lxx = 2Q, luu = 2R,
    lx = 2Qx - 2 x_nom^T Q, Quu not symmetric, Vx' = Qx - Qu^T Quu^{-1} Qux, Vxx' = Qxx - Qux^T Quu^{-1} Qux.  The mid-size
    matrix-core pass (mid_backward, n <= 32) follows it with every use of symmetry switched off (until round 5:
    MI_ILQR_E_UNSUPPORTED): a backward pass on the device's own first trajectory against the NumPy oracle (stage level, 1e-10)
    and a short solve (iterations, step sizes and trial counts exact - or the reference's own line-search failure, which an
    asymmetric Q can cause: its lx is not the gradient of its cost).  Q, R AND Qf asymmetric, x_nom != 0.
    (The reference's recursion is itself fragile on such costs: the antisymmetric part of Vxx enters the symmetric part of Vxx'
    through -Qux^T Quu^{-1} Qux and takes it indefinite within a few dozen steps - on the arm + ball problem the fp64 NumPy pass
    is 1e-12 from the extended-precision one at N = 16, 1e-9 at N = 24 and has no digit left at the scripts' N = 50 (cond(Quu)
    1e14), whatever the size of the asymmetry.  Hence the arm's short horizons here, and the extended-precision yardstick.)
    Round 6: the n = 33 .. 40 kernels too (they refused such matrices until then: their matrix-core chain mirrors tiles) - the
    36-state chain, the 3-D quadruped (n = 37) and plugin chains of (40, 16), (36, 7: padded controls), (33, 16), (40, 8: the
    compact tile layout) through large_backward_asym, the reference's recursion in plain arithmetic."""
    import models as PM
    import plugin_steps as PS
    from drake_ddp_amd import workloads as W
    from drake_ddp_amd.ilqr import BatchedIterativeLQR
    from drake_ddp_amd.models import ModelSystem
    from oracle import models_np as M
    from oracle.ilqr_np import OracleILQR, LinesearchFailed
    rng = np.random.default_rng(23)
    B, cap = 2, 5
    from common import backward_errors
    if not which.startswith("chainx"):
        name, Nw = which.split("_n")
        prob, x0f, ugf = {"arm27": (W.arm27_problem, W.arm27_batch_x0, W.arm27_u_guess), "synth36": (W.synth36_problem, W.synth36_batch_x0, W.synth36_u_guess),
                          "quad3d": (W.quad3d_problem, W.quad3d_batch_x0, W.quad3d_u_guess)}[name]
        prob = prob(N=int(Nw))
        n, m, N, dt = prob["Q"].shape[0], prob["R"].shape[0], prob["N"], prob["dt"]
        sys_ = ModelSystem(prob["model_id"], dt, prob.get("params"))
        model = M.Model(prob["model_id"], dt, prob.get("params"))
        Q, R, Qf, x_nom = prob["Q"].copy(), prob["R"].copy(), prob["Qf"].copy(), prob["x_nom"]
        if name != "arm27":            # (diagonals with zeros: give the asymmetric parts something to scale with)
            Q, Qf = Q + dt * 0.05 * np.eye(n), Qf + 0.05 * np.eye(n)
        x0 = x0f(B)
        ug = np.broadcast_to(ugf(N), (B, m, N - 1)).copy()
        delta, beta = prob["delta"], prob["beta"]
    else:
        nq, m, ne = {"chainx_12_4": (6, 4, 0), "chainx_27_7": (10, 7, 7), "chainx_40_16": (20, 16, 0), "chainx_36_7": (18, 7, 0),
                     "chainx_33_16": (16, 16, 1), "chainx_40_8": (20, 8, 0)}[which]
        n, N, dt = 2 * nq + ne, 24, 0.02
        sys_ = PM.build_chainx(nq, m, ne)(dt)
        model = M.Model.custom(n, m, PS.chainx_step(nq, m, ne), sys_.params, dt)
        Q = dt * np.diag(10.0 ** rng.uniform(-1, 0.5, n)); R = dt * 0.05 * np.eye(m); Qf = np.diag(10.0 ** rng.uniform(0, 1, n))
        x_nom = 0.2 * rng.standard_normal(n)
        x0 = 0.4 * rng.standard_normal((B, n))
        ug = 0.2 * rng.standard_normal((B, m, N - 1))
        delta, beta = 1e-3, 0.7
    # asymmetric parts: strictly upper / lower triangles, 5 - 10 % of sqrt(A_ii A_jj)
    def asym(A, upper):
        d = np.sqrt(np.abs(np.diag(A)))
        T = rng.uniform(0.5, 1.0, A.shape)
        return A + 0.1 * (d[:, None] * (np.triu(T, 1) if upper else np.tril(T, -1)) * d[None, :])
    Q, R, Qf = asym(Q, True), asym(R, False), asym(Qf, False)
    assert not np.allclose(Q, Q.T) and not np.allclose(R, R.T) and not np.allclose(Qf, Qf.T)

    def device(max_iters=None):
        s = BatchedIterativeLQR(sys_, N, B, delta=delta, beta=beta, gamma=0.0, jacobian_mode="ad", on_indefinite="continue",
                                **({"max_iters": max_iters} if max_iters else {}))
        s.SetTargetState(x_nom); s.SetRunningCost(Q, R); s.SetTerminalCost(Qf)
        s.SetInitialState(x0); s.SetInitialGuess(ug)
        return s

    s = device()
    s.stage_forward(np.inf)
    s.stage_backward()
    xb, ub, fx, fu, K1, k1, dV1 = s.x_bar, s.u_bar, s.fx, s.fu, s.K, s.kappa, s.dV_coeff
    worst = 0.0
    for b in range(B):
        o = OracleILQR(model, N, delta, beta, 0.0)
        o.set_problem(x0[b], x_nom, Q, R, Qf, ug[b])
        o.x_bar, o.u_bar, o.fx, o.fu = xb[b], ub[b], fx[b], fu[b]
        o.backward()
        e_dev, e_ref, cond = backward_errors((K1[b], k1[b], dV1[b]), o)
        worst = max(worst, e_dev)
        per = [float(np.max(np.abs(a_ - b_))) / float(np.max(np.abs(b_))) for a_, b_ in ((K1[b], o.K), (k1[b], o.kappa), (dV1[b], o.dV))]
        print(f"{which} problem {b}: device vs NumPy K {per[0]:.1e} kappa {per[1]:.1e} dV {per[2]:.1e}; vs extended {e_dev:.1e} (NumPy {e_ref:.1e}), cond {cond:.1e}, status {int(s.status[b])}")
        assert e_ref < 1e-7, (which, b, e_ref, cond)              # (the comparison means something: the reference has digits here)
        assert e_dev < max(1e-10, 20 * e_ref), (which, b, e_dev, e_ref, cond)
        # (and it is NOT what a symmetrized cost would give: the asymmetric parts are seen)
        osym = OracleILQR(model, N, delta, beta, 0.0)
        osym.set_problem(x0[b], x_nom, 0.5 * (Q + Q.T), 0.5 * (R + R.T), 0.5 * (Qf + Qf.T), ug[b])
        osym.x_bar, osym.u_bar, osym.fx, osym.fu = xb[b], ub[b], fx[b], fu[b]
        osym.backward()
        assert float(np.max(np.abs(osym.K - o.K))) / float(np.max(np.abs(o.K))) > 1e-6
    s = device(cap)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        x, u, _, L = s.Solve()
    n_eps = int(np.floor(np.log(1e-8) / np.log(beta))) + 1
    outcomes = []
    for b in range(B):
        o = OracleILQR(model, N, delta, beta, 0.0, max_iters=cap)
        o.set_problem(x0[b], x_nom, Q, R, Qf, ug[b])
        Lo, done, failed, trials, hist = np.inf, 0, False, 0, []
        while done < cap:
            try:
                L_new, eps, ls = o.forward(Lo)
            except LinesearchFailed:
                failed = True
                break
            o.backward()
            hist.append((L_new, eps, ls)); trials += ls; done += 1
            improvement, Lo = Lo - L_new, L_new
            if not improvement > delta:
                break
        st = int(s.status[b]) & 15
        assert s.iterations[b] == done, (which, b, s.iterations[b], done, st)
        h = s.history[b][:done]
        assert np.array_equal(h[:, 1:3], np.array([[r[1], r[2]] for r in hist]).reshape(done, 2)), (which, b)
        if failed:
            assert st == 2 and s.ls_trials[b] == trials + n_eps
        else:
            assert s.ls_trials[b] == trials and abs(L[b] - Lo) < 1e-9 * abs(Lo)
            assert np.max(np.abs(x[b] - o.x_bar)) < 1e-8 * max(1.0, float(np.abs(o.x_bar).max()))
        outcomes.append("failed" if failed else done)
    print(f"{which}: stage level, worst distance from the extended-precision pass {worst:.1e}; solves (iterations or failed) {outcomes}")
