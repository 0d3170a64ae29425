"""Shape sweep of the plugin interface against the NumPy oracle: chain models of many (n, m) on both kernel families, two iLQR
iterations each with forward-mode duals and with central differences - x_bar, K, kappa against oracle/ilqr_np.py.  The sweep
that found the wrong gains of (36, 4), (36, 8), (40, 4) and m = 16 above 32 states in round 4.
    python tools/shape_sweep.py build      compile the plugins (CPU, in parallel)
    python tools/shape_sweep.py            run (GPU)
MI_ILQR_CLUSTER=k forces clusters of k workgroups per problem onto every workgroup-per-problem shape (plugin models are not clustered by
default): the sweep that, in round 5, showed where the early rounds of ilqr_large.hpp are not safe."""
import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "examples", "plugins"))
import models as PM
import plugin_steps as PS
from drake_ddp_amd import plugin
from drake_ddp_amd.ilqr import BatchedIterativeLQR
from oracle import models_np as M
from oracle.ilqr_np import OracleILQR

rng0 = np.random.default_rng(2026)
large = []
for n in (2, 3, 4, 5, 7, 8, 9, 11, 13, 15, 16, 17, 18, 19, 21, 23, 24, 25, 29, 31, 32, 33, 35, 36, 38, 40):
    for m in sorted(set(int(v) for v in rng0.choice([1, 2, 3, 5, 6, 9, 10, 11, 13, 14, 15, 16], 2, replace=False))):
        large.append((n // 2, m, n % 2, "large"))
small = [(n // 2, m, n % 2, "small") for n in (2, 3, 4, 5, 6, 7, 8, 9, 10, 12) for m in (1, 2)]
shapes = large + small


def spec(nq, m, ne, fam):
    name, n, m_, body, defaults, _ = PM.chainx_spec(nq, m, ne)
    return (name + "_" + fam, n, m_, body, defaults, fam)


if len(sys.argv) > 1 and sys.argv[1] == "build":
    import concurrent.futures
    with concurrent.futures.ThreadPoolExecutor(os.cpu_count() or 4) as ex:
        list(ex.map(lambda sp: plugin.compile_model(*sp), [spec(*sh) for sh in shapes]))
    sys.exit(0)
if len(sys.argv) == 1:                                   # chunks of 24 in child processes (one hipModule set per chunk keeps the process small)
    import subprocess
    rc = 0
    for c in range((len(shapes) + 23) // 24):
        rc |= subprocess.call([sys.executable, os.path.abspath(__file__), str(c)])
    sys.exit(rc)
shapes = shapes[24 * int(sys.argv[1]):24 * int(sys.argv[1]) + 24]
make = plugin.build_models([spec(*sh) for sh in shapes], verbose=False)
bad = 0
for nq, m, ne, fam in shapes:
    n = 2 * nq + ne
    dt, B = 0.02, int(os.environ.get("SWEEP_B", "2"))
    sys_ = make["chainx_%d_%d_%d_%s" % (nq, m, ne, fam)](dt)
    model = M.Model.custom(n, m, PS.chainx_step(nq, m, ne), sys_.params, dt)
    for N in [int(v) for v in os.environ.get("SWEEP_N", "24,4").split(",")]:
        rng = np.random.default_rng(n * 17 + m)
        x_nom = np.zeros(n); x0 = 0.4 * rng.standard_normal((B, n)); ug = 0.2 * rng.standard_normal((B, m, N - 1))
        Q = dt * np.diag(10.0 ** rng.uniform(-1, 0.5, n)); R = dt * 0.05 * np.eye(m); Qf = np.diag(10.0 ** rng.uniform(0, 1, n))
        for jac in ("ad", "fd"):
            try:
                s = BatchedIterativeLQR(sys_, N, B, delta=1e-3, beta=0.7, gamma=0.0, jacobian_mode=jac, max_iters=2)
            except Exception as e:
                print(f"{fam} n={n} m={m} N={N} {jac}: create refused: {e}"); bad += 1; continue
            s.SetTargetState(x_nom); s.SetRunningCost(Q, R); s.SetTerminalCost(Qf); s.SetInitialState(x0); s.SetInitialGuess(ug)
            try:
                s.Solve()
            except RuntimeError:
                pass
            o = OracleILQR(model, N, 1e-3, 0.7, 0.0, jacobian=jac, fd_step=1e-5, max_iters=2)
            o.set_problem(x0[1], x_nom, Q, R, Qf, ug[1]); xo, uo, Lo, hist = o.solve()
            ek = np.abs(s.K[1] - o.K).max() / max(1.0, np.abs(o.K).max()); ex = np.abs(s.x_bar[1] - xo).max()
            ekap = np.abs(s.kappa[1] - o.kappa).max() / max(1.0, np.abs(o.kappa).max())
            tol = 1e-9 if jac == "ad" else 1e-5
            ok = ek < tol and ex < tol and ekap < tol and s.iterations[1] == len(hist)
            bad += not ok
            if not ok or (N == 24 and jac == "ad"):
                print(f"{fam:5s} n={n:2d} m={m:2d} N={N:2d} {jac}: {'ok ' if ok else 'BAD'} x {ex:.1e} K {ek:.1e} kappa {ekap:.1e}", flush=True)
if os.environ.get("SWEEP_MPC"):
    # converged solve + two receding-horizon re-solves (device loop) + an adaptive-jerk key-point solve, per shape
    from drake_ddp_amd import utils_derivs_interpolation as U
    from drake_ddp_amd.workloads import mpc_shift
    from oracle.ilqr_np import KeypointCfg
    for nq, m, ne, fam in shapes:
        n = 2 * nq + ne
        dt, B, N = 0.02, 2, 16
        sys_ = make["chainx_%d_%d_%d_%s" % (nq, m, ne, fam)](dt)
        model = M.Model.custom(n, m, PS.chainx_step(nq, m, ne), sys_.params, dt)
        rng = np.random.default_rng(n * 31 + m)
        x_nom = np.zeros(n); x0 = 0.3 * rng.standard_normal((B, n)); ug = 0.1 * rng.standard_normal((m, N - 1))
        Q, R, Qf = dt * np.eye(n), dt * 0.05 * np.eye(m), 5.0 * np.eye(n)
        s = BatchedIterativeLQR(sys_, N, B, delta=1e-3, beta=0.6, gamma=0.0, jacobian_mode="ad")
        s.SetTargetState(x_nom); s.SetRunningCost(Q, R); s.SetTerminalCost(Qf); s.SetInitialState(x0); s.SetInitialGuess(ug)
        s.Solve(); it0 = s.iterations.copy()
        s.MPCRun(2, 3); log = s.mpc_log
        o = OracleILQR(model, N, 1e-3, 0.6, 0.0, jacobian="ad")
        o.set_problem(x0[1], x_nom, Q, R, Qf, ug); xo, uo, Lo, hist = o.solve()
        ok = len(hist) == it0[1]
        for r_ in range(2):
            x0r, ugr = mpc_shift(xo, uo, 3)
            o.set_problem(x0r, x_nom, Q, R, Qf, ugr); xo, uo, Lo, hist = o.solve()
            ok = ok and log[1, r_, -1] == len(hist) and abs(log[1, r_, -2] - Lo) < 1e-8 * abs(Lo)
        kp = ("adaptiveJerk", 2, 5, 1e-3, 0.0)
        s2 = BatchedIterativeLQR(sys_, N, B, delta=1e-3, beta=0.6, gamma=0.0, jacobian_mode="ad", derivs_keypoint_method=U.derivs_interpolation(*kp))
        s2.SetTargetState(x_nom); s2.SetRunningCost(Q, R); s2.SetTerminalCost(Qf); s2.SetInitialState(x0); s2.SetInitialGuess(ug)
        s2.Solve()
        o2 = OracleILQR(model, N, 1e-3, 0.6, 0.0, keypoint=KeypointCfg(*kp), jacobian="ad")
        o2.set_problem(x0[1], x_nom, Q, R, Qf, ug)
        nk = int(s2.keypoint_count[1])
        try:
            xo2, uo2, Lo2, hist2 = o2.solve()
            ok2 = len(hist2) == s2.iterations[1] and list(s2.keypoint_list[1][:nk]) == list(o2.keypoints) and abs(s2.cost[1] - Lo2) < 1e-8 * abs(Lo2)
        except Exception:                      # (interpolated Jacobians: the reference's line search may run out of step sizes, ilqr.py:337)
            # ... or, at the optimum, tie on the last bit of L_last - L > 0: the device then either fails the same way or accepts a
            # step of ~1e-8 that changes nothing and converges
            h2 = s2.history[1]
            it2 = int(s2.iterations[1])
            ok2 = int(s2.status[1]) == 2 or (it2 >= 2 and abs(h2[it2 - 1, 0] - h2[it2 - 2, 0]) <= 1e-9 * abs(h2[it2 - 1, 0]))
        bad += (not ok) + (not ok2)
        if not (ok and ok2):
            print(f"{fam:5s} n={n:2d} m={m:2d}: MPC {'ok' if ok else 'BAD'} key-points {'ok' if ok2 else 'BAD'}", flush=True)
print("shapes", len(shapes), "bad", bad)
sys.exit(1 if bad else 0)
