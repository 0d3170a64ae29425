"""Per-stage cycles of the arm + ball receding-horizon configs (C6, C6b: B = 64, clusters of 4, 1 + 20 solves): line search per
trial and per iteration, linearization, backward pass (in-kernel stopwatches of the MPC launch), and how the early rounds fared."""
import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from drake_ddp_amd import workloads as W
from test_gpu_parity import make_solver

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
for name, p, ug in (("C6 arm27", W.arm27_problem(), W.arm27_u_guess(50)), ("C6b arm27c", W.arm27c_problem(), W.arm27c_u_guess(50))):
    N = p["N"]
    s = make_solver(p, B=B, jac="fd")
    s.SetInitialState(W.arm27_batch_x0(B)); s.SetInitialGuess(ug)
    s.Solve()
    ls0 = s.ls_trials.copy()
    s.MPCRun(20, 5)
    cyc = s.stage_cycles.astype(float); it = s.iterations.astype(float); ls = s.ls_trials.astype(float) - ls0
    cs = s.cluster_stats
    print(f"{name:11s} B {B}: MPC launch {s.stats.kernel_ms:7.2f} ms, iterations {int(it.sum())} (max {int(it.max())}, min {int(it.min())}), trials {int(ls.sum())} | cycles per iteration: "
          f"line search {(cyc[:, 0] / it).mean():8.0f} linearize {(cyc[:, 1] / it).mean():8.0f} backward {(cyc[:, 2] / it).mean():8.0f} (per step {(cyc[:, 2] / it).mean() / (N - 1):6.0f}) "
          f"all {(cyc[:, 3] / it).mean():8.0f} | loop cycles max {cyc[:, 3].max():.0f} mean {cyc[:, 3].mean():.0f} | regular rounds {cs[:, 1].sum()} early opened {cs[:, 3].sum()} hit {cs[:, 4].sum()} candidate-group rounds {cs[:, 5].sum()} | reference trials in the loop {int(s.ls_trials.sum())}", flush=True)
