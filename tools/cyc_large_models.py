"""Per-stage cycles of the workgroup-per-problem kernels (planar quadruped, 3-D quadruped, synthetic chain) at B = 64 and 8:
line search, linearization, backward pass per iteration (in-kernel stopwatches)."""
import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from drake_ddp_amd import workloads as W
from test_gpu_parity import make_solver
for name, prob, x0, ug in (("quad", W.planar_quad_problem(), W.planar_quad_batch_x0(64), W.planar_quad_u_guess(40)),
                           ("quad3d", W.quad3d_problem(), W.quad3d_batch_x0(64), W.quad3d_u_guess(W.quad3d_problem()["N"])),
                           ("synth36", W.synth36_problem(), W.synth36_batch_x0(64), W.synth36_u_guess(40))):
    for B in (64, 8):
        s = make_solver(prob, B=B, jac="fd")
        s.SetInitialState(x0[:B]); s.SetInitialGuess(ug)
        s.Solve(); s.Reset(); s.SetInitialGuess(ug); s.Solve()
        cyc = s.stage_cycles.astype(float); it = s.iterations; ls = s.ls_trials
        print(name, "B", B, "kernel_ms", round(s.stats.kernel_ms, 3), "iters mean", it.mean(), "ls mean", ls.mean(),
              "per-iteration cycles: linesearch", round((cyc[:, 0] / it).mean()), "per trial", round((cyc[:, 0] / ls).mean()),
              "linearize", round((cyc[:, 1] / it).mean()), "backward", round((cyc[:, 2] / it).mean()), "total/iter", round((cyc[:, 3] / it).mean()))
