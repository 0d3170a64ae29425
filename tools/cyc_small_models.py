"""Per-stage cycles per iteration of the wave-per-problem kernels on C3 (acrobot MPC) and C4 (cart-pole with wall)."""
import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from drake_ddp_amd import workloads as W
from test_gpu_parity import make_solver
a = W.acrobot_problem()
s = make_solver(a, B=512, jac="fd")
s.SetInitialState(W.acrobot_batch_x0(512)); s.SetInitialGuess(np.zeros((1, a["N"] - 1)))
s.Solve()
cyc = s.stage_cycles.astype(float); it = s.iterations; ls = s.ls_trials
print("C3 cold solve: kernel_ms", round(s.stats.kernel_ms, 4), "iters mean/max", it.mean(), it.max(), "ls/iter", (ls / it).mean(),
      "cycles/iter: ls", round((cyc[:, 0] / it).mean()), "lin", round((cyc[:, 1] / it).mean()), "bp", round((cyc[:, 2] / it).mean()), "total", round((cyc[:, 3] / it).mean()))
st = s.MPCRun(50, 2)
cyc = s.stage_cycles.astype(float); it = s.iterations; ls = s.ls_trials
print("C3 MPC x50: kernel_ms", round(st.kernel_ms, 4), "iters mean/max", it.mean(), it.max(), "ls/iter", (ls / it).mean(),
      "cycles/iter: ls", round((cyc[:, 0] / it).mean()), "lin", round((cyc[:, 1] / it).mean()), "bp", round((cyc[:, 2] / it).mean()), "total", round((cyc[:, 3] / it).mean()),
      "slowest problem total cycles", cyc[:, 3].max(), "mean", cyc[:, 3].mean())
c = W.cartpole_wall_problem()
s = make_solver(c, B=256, jac="fd", hist_cap=8)
s.SetInitialState(W.cartpole_wall_batch_x0(256)); s.SetInitialGuess(np.zeros((1, c["N"] - 1)))
s.Solve()
cyc = s.stage_cycles.astype(float); it = s.iterations; ls = s.ls_trials
print("C4: kernel_ms", round(s.stats.kernel_ms, 4), "iters mean/max", it.mean(), it.max(), "ls/iter", (ls / it).mean(),
      "cycles/iter: ls", round((cyc[:, 0] / it).mean()), "lin", round((cyc[:, 1] / it).mean()), "bp", round((cyc[:, 2] / it).mean()), "total", round((cyc[:, 3] / it).mean()))
