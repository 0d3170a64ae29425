"""Per-stage cycles of the mid-size workgroup-per-problem kernels (n <= 32): the arm + ball model (n = 27, m = 7) and plugin
chains of other shapes, at several batch sizes - line search (per trial), linearization, backward pass per iteration
(in-kernel stopwatches), and per backward STEP."""
import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "examples", "plugins"))
from drake_ddp_amd import workloads as W
from drake_ddp_amd.ilqr import BatchedIterativeLQR
from test_gpu_parity import make_solver
import models as PM


def report(name, s, N):
    cyc = s.stage_cycles.astype(float); it = s.iterations; ls = s.ls_trials
    print(f"{name:28s} B {s.B:5d} kernel_ms {s.stats.kernel_ms:8.3f} iters mean {it.mean():5.1f} max {it.max():3d} trials mean {ls.mean():5.1f} | cycles: per trial "
          f"{(cyc[:, 0] / ls).mean():8.0f} (per step {(cyc[:, 0] / ls).mean() / (N - 1):6.0f}) linearize {(cyc[:, 1] / it).mean():8.0f} backward {(cyc[:, 2] / it).mean():8.0f} "
          f"(per step {(cyc[:, 2] / it).mean() / (N - 1):6.0f}) iteration {(cyc[:, 3] / it).mean():8.0f} | {it.sum() / (s.stats.kernel_ms * 1e-3):10.0f} it/s", flush=True)


jacs = sys.argv[1:] or ["fd"]
for jac in jacs:
    p = W.arm27_problem()
    for B in (1, 64, 256, 1024):
        s = make_solver(p, B=B, jac=jac)
        s.SetInitialState(W.arm27_batch_x0(B)); s.SetInitialGuess(W.arm27_u_guess(p["N"]))
        s.Solve(); s.Reset(); s.SetInitialGuess(W.arm27_u_guess(p["N"])); s.Solve()
        report(f"arm27 {jac}", s, p["N"])
    pc = W.arm27c_problem()
    for B in (1, 64, 1024):
        s = make_solver(pc, B=B, jac=jac)
        s.SetInitialState(W.arm27_batch_x0(B)); s.SetInitialGuess(W.arm27c_u_guess(pc["N"]))
        s.Solve(); s.Reset(); s.SetInitialGuess(W.arm27c_u_guess(pc["N"])); s.Solve()
        report(f"arm27c {jac}", s, pc["N"])
    if os.environ.get("MI_CYC_ARMS_ONLY") == "1":
        continue
    for nq, m, ne in ((6, 4, 0), (7, 7, 0), (16, 16, 0), (8, 1, 0)):
        n = 2 * nq + ne
        dt, N = 0.02, 50
        sys_ = PM.build_chainx(nq, m, ne)(dt)
        rng = np.random.default_rng(0)
        for B in (64, 1024):
            s = BatchedIterativeLQR(sys_, N, B, delta=1e-3, beta=0.7, jacobian_mode=jac)
            s.SetTargetState(np.zeros(n)); s.SetRunningCost(dt * np.eye(n), dt * 0.05 * np.eye(m)); s.SetTerminalCost(5.0 * np.eye(n))
            s.SetInitialState(0.4 * rng.standard_normal((B, n))); s.SetInitialGuess(np.zeros((m, N - 1)))
            s.Solve(); s.Reset(); s.SetInitialGuess(np.zeros((m, N - 1))); s.Solve()
            report(f"chainx n={n} m={m} {jac}", s, N)
