"""The open model interface (include/mi_ilqr.h: mi_ilqr_register_model; drake_ddp_amd/plugin.py): models that are not
compiled into libmi_ilqr.so, built as plugins from the C++ body of their discrete update (examples/plugins/models.py) and
checked against the NumPy oracle driven by the same update written in Python (the reference takes ANY discrete System:
ilqr.py:21,37-58)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "examples", "plugins"))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def test_plugin_registers_without_a_gpu():
    """Built by __graft_entry__.build(); loading and registering needs no device: ids from MI_MODEL_PLUGIN_BASE, the
    registry answers mi_ilqr_model_info for them, bad records are refused."""
    import ctypes as C
    import models as PM
    from drake_ddp_amd import _capi, plugin
    make = PM.build_all()
    lib = _capi.load()
    for name, n, m, defaults in (("vdp", 2, 1, PM.VDP_DEFAULTS), ("chain3", 6, 2, PM.CHAIN3_DEFAULTS), ("synth36p", 36, 12, PM.SYNTH36P_DEFAULTS)):
        sys_ = make[name](0.01)
        assert sys_.model_id >= 100 and (sys_.n, sys_.m) == (n, m) and list(sys_.params) == defaults
        nn, mm, npar = C.c_int32(), C.c_int32(), C.c_int32()
        dp = (C.c_double * _capi.MAX_PARAMS)()
        assert lib.mi_ilqr_model_info(sys_.model_id, C.byref(nn), C.byref(mm), C.byref(npar), dp) == 0
        assert (nn.value, mm.value, npar.value) == (n, m, len(defaults)) and list(dp[:len(defaults)]) == defaults
    assert make["vdp"](0.01).model_id == make["vdp"](0.02).model_id                  # registered once per process
    good = plugin._Plugin()
    plug = C.CDLL(plugin.plugin_path("vdp", plugin.source("vdp", 2, 1, PM.VDP_BODY, PM.VDP_DEFAULTS)))
    plug.mi_plugin_describe.argtypes = [C.POINTER(plugin._Plugin)]
    plug.mi_plugin_describe(C.byref(good))
    assert good.abi_version == _capi.ABI_VERSION and good.kernel_args_bytes > 500
    mid = C.c_int32()
    bad = plugin._Plugin(abi_version=good.abi_version, kernel_args_bytes=good.kernel_args_bytes, handle_bytes=good.handle_bytes, n=2, m=3, n_params=0, family=0, launch=1, lds_bytes=1)
    assert lib.mi_ilqr_register_model(C.byref(bad), C.byref(mid)) == _capi.E_UNSUPPORTED     # m > 2 on the wave-per-problem family
    stale = plugin._Plugin(abi_version=good.abi_version, kernel_args_bytes=good.kernel_args_bytes - 8, handle_bytes=good.handle_bytes, n=2, m=1, n_params=0, family=0, launch=1, lds_bytes=1)
    assert lib.mi_ilqr_register_model(C.byref(stale), C.byref(mid)) == _capi.E_BAD_ARG       # built against other headers
    stale2 = plugin._Plugin(abi_version=good.abi_version, kernel_args_bytes=good.kernel_args_bytes, handle_bytes=good.handle_bytes + 8, n=2, m=1, n_params=0, family=0, launch=1, lds_bytes=1)
    assert lib.mi_ilqr_register_model(C.byref(stale2), C.byref(mid)) == _capi.E_BAD_ARG      # ... with another handle layout
    src = plugin.source("vdp", 2, 1, PM.VDP_BODY, PM.VDP_DEFAULTS)
    assert "launch_small.hpp" in src and "mi_plugin_describe" in src and "PluginModel" in src


CASES = {
    "vdp": dict(n=2, m=1, dt=0.02, N=100, x_nom=np.zeros(2), Q=np.eye(2), R=0.1 * np.eye(1), Qf=10.0 * np.eye(2), span=2.0),
    # (target behind the wall at q = 0, starts on both sides of it: every trajectory crosses the kink, many inside a lane's chunk
    #  of steps; gamma = 0.1 exercises the expected-improvement term of the acceptance test, ilqr.py:330-331.  The model keeps
    #  a smooth nonlinearity beside the kink on purpose: with piecewise-LINEAR dynamics iLQR lands exactly on a stationary
    #  point, and whether its next iteration's 1e-16 "improvement" counts as a decrease or ends in "linesearch failed" is
    #  round-off on either side - observed in the NumPy oracle and on the device alike)
    "kink2": dict(n=2, m=1, dt=0.01, N=180, x_nom=np.array([-0.05, 0.0]), Q=np.eye(2), R=0.02 * np.eye(1), Qf=20.0 * np.eye(2), span=0.6, gamma=0.1),
    "chain3": dict(n=6, m=2, dt=0.02, N=60, x_nom=np.array([np.pi, np.pi, np.pi, 0, 0, 0.0]), Q=np.diag([1, 1, 1, .1, .1, .1]),
                   R=0.05 * np.eye(2), Qf=20.0 * np.eye(6), span=0.6),
}


@pytest.mark.gpu
@pytest.mark.parametrize("jac", ["ad", "fd"])
@pytest.mark.parametrize("name", ["vdp", "chain3", "kink2"])
def test_plugin_model_solves_like_the_oracle(name, jac):
    """A batch of 24 problems on a plugin model: iterations and line-search trials of every problem exactly the NumPy
    oracle's (driven by the Python statement of the same update), costs 1e-9 (duals) / 1e-8 (central differences),
    trajectories 1e-6; then three receding-horizon re-solves on the device (mi_ilqr_mpc_run) against the oracle's loop."""
    import models as PM
    from drake_ddp_amd.ilqr import BatchedIterativeLQR
    from drake_ddp_amd.workloads import mpc_shift
    from oracle import models_np as M
    from oracle.ilqr_np import OracleILQR
    c = CASES[name]
    n, m, dt, N = c["n"], c["m"], c["dt"], c["N"]
    sys_ = PM.build_all()[name](dt)
    B = 24
    rng = np.random.default_rng(11)
    x0 = c["x_nom"] + rng.uniform(-c["span"], c["span"], (B, n))
    ug = rng.uniform(-0.1, 0.1, (m, N - 1))
    gamma = c.get("gamma", 0.0)
    s = BatchedIterativeLQR(sys_, N, B, delta=1e-3, beta=0.7, gamma=gamma, jacobian_mode=jac, hist_cap=64)
    s.SetTargetState(c["x_nom"]); s.SetRunningCost(dt * c["Q"], dt * c["R"]); s.SetTerminalCost(c["Qf"])
    s.SetInitialState(x0); s.SetInitialGuess(ug)
    x, u, _, L = s.Solve()
    assert (s.status == 0).all()
    import plugin_steps as PS
    step_fn = {"vdp": PS.vdp_step, "chain3": PS.chain3_step, "kink2": PS.kink2_step}[name]
    model = M.Model.custom(n, m, step_fn, sys_.params, dt)
    tolL = 1e-9 if jac == "ad" else 1e-8
    oracles = []
    for b in range(B):
        o = OracleILQR(model, N, 1e-3, 0.7, gamma, jacobian=jac, fd_step=1e-5)
        o.set_problem(x0[b], c["x_nom"], dt * c["Q"], dt * c["R"], c["Qf"], ug)
        xo, uo, Lo, hist = o.solve()
        hist = np.array(hist)
        assert len(hist) == s.iterations[b] and int(hist[:, 2].sum()) == s.ls_trials[b], b
        assert np.array_equal(s.history[b][:len(hist), 1:3], hist[:, 1:3]), b
        assert abs(L[b] - Lo) < tolL * abs(Lo) and np.max(np.abs(x[b] - xo)) < 1e-6 * max(1.0, np.abs(xo).max()), b
        oracles.append((o, xo, uo))
    s.MPCRun(3, 2)
    log = s.mpc_log
    for b in range(0, B, 6):
        o, xo, uo = oracles[b]
        for r in range(3):
            x0r, ugr = mpc_shift(xo, uo, 2)
            o.set_problem(x0r, c["x_nom"], dt * c["Q"], dt * c["R"], c["Qf"], ugr)
            xo, uo, Lo, hist = o.solve()
            assert log[b, r, -1] == len(hist) and abs(log[b, r, -2] - Lo) < 10 * tolL * abs(Lo), (b, r)


@pytest.mark.gpu
def test_plugin_on_the_matrix_core_family_solves_like_the_builtin_model():
    """Family 1 of the open interface (workgroup-per-problem kernels, 32 < n <= 40): the built-in 36-state chain written
    again as a plugin in whole-step form - one lane advances the dynamics in the rollout, dense whole-step Jacobian columns -
    must reproduce the built-in model's solve and MPC loop (same formulas, other evaluation order of the linearization:
    counts exact, costs 1e-9, trajectories 1e-8)."""
    import models as PM
    from drake_ddp_amd import workloads as W
    from drake_ddp_amd.ilqr import BatchedIterativeLQR
    from drake_ddp_amd.models import Synth36
    q = W.synth36_problem()
    B = 6
    x0, ug = W.synth36_batch_x0(64)[:B], W.synth36_u_guess(q["N"])
    step = np.zeros(36)
    step[0] = W.SYNTH_TARGET_VEL * q["dt"] * 4
    res = []
    for sys_ in (Synth36(q["dt"]), PM.build_all()["synth36p"](q["dt"])):
        s = BatchedIterativeLQR(sys_, q["N"], B, delta=q["delta"], beta=q["beta"], gamma=q["gamma"], jacobian_mode="fd")
        s.SetTargetState(q["x_nom"]); s.SetRunningCost(q["Q"], q["R"]); s.SetTerminalCost(q["Qf"])
        s.SetInitialState(x0); s.SetInitialGuess(ug)
        x, u, _, L = s.Solve()
        it0 = s.iterations.copy()
        s.MPCRun(5, 4, target_step=step)
        res.append((x, L, it0, s.mpc_log, s.x_bar, s.status))
    a, b = res
    assert (a[5] == 0).all() and (b[5] == 0).all()
    assert np.array_equal(a[2], b[2]) and np.array_equal(a[3][:, :, -1], b[3][:, :, -1])
    assert np.max(np.abs(a[1] - b[1]) / np.abs(a[1])) < 1e-9 and np.max(np.abs(a[0] - b[0])) < 1e-8
    assert np.max(np.abs(a[3][:, :, -2] - b[3][:, :, -2]) / np.abs(a[3][:, :, -2])) < 1e-8 and np.max(np.abs(a[4] - b[4])) < 1e-7


@pytest.mark.gpu
@pytest.mark.parametrize("nq", [17, 20])
def test_matrix_core_family_at_other_state_dimensions(nq):
    """Family 1 beyond the two shapes the library ships: chains of 17 and 20 pendula (n = 34, 40; m = 12) - the split
    tile layout with one resp. two four-row groups in the thin last row tile of the backward pass (n = 34: a state
    dimension that is not a multiple of four; n = 40: the largest the by-value kernel arguments admit).  Forward-mode duals
    against the NumPy oracle driven by the Python statement of the same update, after two iterations (round-off) and at
    convergence (cost)."""
    import models as PM
    from drake_ddp_amd.ilqr import BatchedIterativeLQR
    from oracle import models_np as M
    from oracle.ilqr_np import OracleILQR
    n, m, dt, N, B = 2 * nq, 12, 0.02, 24, 3
    sys_ = PM.build_chain(nq)(dt)
    rng = np.random.default_rng(nq)
    x_nom = np.zeros(n)
    x0 = 0.4 * rng.standard_normal((B, n))
    ug = 0.2 * rng.standard_normal((B, m, N - 1))
    Q = dt * np.diag(10.0 ** rng.uniform(-1, 0.5, n))
    R = dt * 0.05 * np.eye(m)
    Qf = np.diag(10.0 ** rng.uniform(0, 1, n))
    import plugin_steps as PS
    model = M.Model.custom(n, m, PS.chain_step(nq), sys_.params, dt)
    for cap in (2, 100000):
        s = BatchedIterativeLQR(sys_, N, B, delta=1e-3, beta=0.7, gamma=0.0, jacobian_mode="ad", **({"max_iters": cap} if cap == 2 else {}))
        s.SetTargetState(x_nom); s.SetRunningCost(Q, R); s.SetTerminalCost(Qf)
        s.SetInitialState(x0); s.SetInitialGuess(ug)
        try:
            s.Solve()
        except RuntimeError:
            assert cap == 2
        for b in range(B):
            o = OracleILQR(model, N, 1e-3, 0.7, 0.0, jacobian="ad", max_iters=cap)
            o.set_problem(x0[b], x_nom, Q, R, Qf, ug[b])
            xo, uo, Lo, hist = o.solve()
            if cap == 2:
                assert s.iterations[b] == len(hist) and [h[1] for h in hist] == list(s.history[b][:len(hist), 1])
                sc = lambda a_: max(1.0, float(np.max(np.abs(a_))))
                assert np.max(np.abs(s.x_bar[b] - xo)) < 1e-11 * sc(xo) and np.max(np.abs(s.K[b] - o.K)) < 1e-10 * sc(o.K)
                assert np.max(np.abs(s.kappa[b] - o.kappa)) < 1e-10 * sc(o.kappa)
            else:
                assert s.status[b] == 0 and abs(s.cost[b] - Lo) < 1e-10 * abs(Lo)


@pytest.mark.gpu
def test_family0_plugins_on_the_lane_per_problem_kernels():
    """Family-0 plugin units (n <= 6) instantiate the lane-per-problem "throughput" kernels as well: kernel_mode = throughput,
    and - under AUTO - horizons whose state does not fit LDS (N = 900 for the n = 6, m = 2 chain: refused until round 4).
    Against the wave-per-problem kernels of the same plugin (decisions identical, costs to 1e-9) and the NumPy oracle; with a
    key-point method and through the receding-horizon loop (host-loop form) as well."""
    import models as PM
    import plugin_steps as PS
    from drake_ddp_amd import utils_derivs_interpolation as U
    from drake_ddp_amd.ilqr import BatchedIterativeLQR
    from oracle import models_np as M
    from oracle.ilqr_np import OracleILQR
    make = PM.build_all()
    rng = np.random.default_rng(4)
    for name, n, m, step, N in (("vdp", 2, 1, PS.vdp_step, 80), ("chain3", 6, 2, PS.chain3_step, 50)):
        dt, B = 0.02, 70
        sys_ = make[name](dt)
        x0 = rng.uniform(-1.0, 1.0, (B, n)); ug = np.zeros((m, N - 1))
        Q, R, Qf = dt * np.eye(n), dt * 0.1 * np.eye(m), 10.0 * np.eye(n)
        out = {}
        for mode in ("latency", "throughput"):
            for kp in (None, U.derivs_interpolation("adaptiveJerk", 2, 8, 1e-4, 0.0)):
                s = BatchedIterativeLQR(sys_, N, B, delta=1e-3, beta=0.7, jacobian_mode="fd", kernel_mode=mode, derivs_keypoint_method=kp)
                s.SetTargetState(np.zeros(n)); s.SetRunningCost(Q, R); s.SetTerminalCost(Qf)
                s.SetInitialState(x0); s.SetInitialGuess(ug)
                s.Solve()
                out[mode, kp is None] = (s.iterations.copy(), s.ls_trials.copy(), s.cost.copy(), s.keypoint_count.copy())
                if mode == "throughput" and kp is None:
                    s.MPCRun(3, 2)
                    assert (s.status == 0).all() and s.mpc_log.shape == (B, 3, n + 2)
        for plain in (True, False):
            a, b = out["latency", plain], out["throughput", plain]
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[3], b[3]), (name, plain)
            assert np.max(np.abs(a[2] - b[2]) / np.abs(a[2])) < 1e-9
        o = OracleILQR(M.Model.custom(n, m, step, sys_.params, dt), N, 1e-3, 0.7, 0.0, jacobian="fd", fd_step=1e-5)
        o.set_problem(x0[0], np.zeros(n), Q, R, Qf, ug)
        xo, uo, Lo, hist = o.solve()
        assert out["throughput", True][0][0] == len(hist) and abs(out["throughput", True][2][0] - Lo) < 1e-8 * abs(Lo)
    # a horizon beyond LDS: AUTO takes the streaming kernels
    n, m, N, dt, B = 6, 2, 900, 0.01, 5
    s = BatchedIterativeLQR(make["chain3"](dt), N, B, delta=1e-3, beta=0.7, jacobian_mode="fd")
    s.SetTargetState(np.zeros(n)); s.SetRunningCost(dt * np.eye(n), dt * 0.1 * np.eye(m)); s.SetTerminalCost(10.0 * np.eye(n))
    x0 = rng.uniform(-0.5, 0.5, (B, n))
    s.SetInitialState(x0); s.SetInitialGuess(np.zeros((m, N - 1)))
    s.Solve()
    o = OracleILQR(M.Model.custom(n, m, PS.chain3_step, make["chain3"](dt).params, dt), N, 1e-3, 0.7, 0.0, jacobian="fd", fd_step=1e-5)
    o.set_problem(x0[1], np.zeros(n), dt * np.eye(n), dt * 0.1 * np.eye(m), 10.0 * np.eye(n), np.zeros((m, N - 1)))
    xo, uo, Lo, hist = o.solve()
    assert (s.status == 0).all() and s.iterations[1] == len(hist) and abs(s.cost[1] - Lo) < 1e-7 * abs(Lo)
