"""Per-stage cycles of the wave-per-problem kernels on the bench's configs: C3 (acrobot MPC B = 512), C1 (single pendulum), C4
(cart-pole + wall B = 256) - line search per trial, linearization, backward pass per iteration (in-kernel stopwatches)."""
import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from drake_ddp_amd import workloads as W
from test_gpu_parity import make_solver


def report(name, s, it, ls, N):
    cyc = s.stage_cycles.astype(float)
    print(f"{name:32s}: kernel {s.stats.kernel_ms:7.3f} ms, iterations {int(it.sum())} (max {int(it.max())}), trials {int(ls.sum())} | cycles per iteration: line search "
          f"{(cyc[:, 0] / it).mean():8.0f} (per trial and step {(cyc[:, 0] / ls).mean() / (N - 1):6.0f}) linearize {(cyc[:, 1] / it).mean():8.0f} backward {(cyc[:, 2] / it).mean():8.0f} "
          f"all {(cyc[:, 3] / it).mean():8.0f}; whole loop of the slowest problem {cyc[:, 3].max():9.0f}", flush=True)


a = W.acrobot_problem()
for B in (512, 64):
    s = make_solver(a, B=B, jac="fd")
    s.SetInitialState(W.acrobot_batch_x0(512)[:B]); s.SetInitialGuess(np.zeros((1, a["N"] - 1)))
    s.Solve()
    report(f"acrobot cold solve B = {B}", s, s.iterations.astype(float), s.ls_trials.astype(float), a["N"])
    it0, ls0 = s.iterations.copy(), s.ls_trials.copy()
    s.MPCRun(50, 2)
    report(f"acrobot MPC 50 re-solves B = {B}", s, s.iterations.astype(float), s.ls_trials.astype(float) - ls0, a["N"])
c = W.cartpole_wall_problem()
s = make_solver(c, B=256, jac="fd")
s.SetInitialState(W.cartpole_wall_batch_x0(256)); s.SetInitialGuess(np.zeros((1, c["N"] - 1)))
s.Solve()
report("cart-pole + wall B = 256", s, s.iterations.astype(float), s.ls_trials.astype(float), c["N"])
p = W.pendulum_problem()
s = make_solver(p, B=1024, jac="fd")
s.SetInitialState(W.pendulum_batch_x0(1024)); s.SetInitialGuess(np.zeros((1, p["N"] - 1)))
s.Solve()
report("pendulum B = 1024", s, s.iterations.astype(float), s.ls_trials.astype(float), p["N"])
