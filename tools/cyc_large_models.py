"""Per-stage cycles of the n = 36 / 37 workgroup-per-problem kernels on the bench's MPC configs (C5, C5q, C5q3d; B = 64, clusters of
4): line search per trial, linearization, backward pass per iteration (in-kernel stopwatches of the MPC launch)."""
import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from drake_ddp_amd import workloads as W
from test_gpu_parity import make_solver

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
for name, q, x0, ug, col, vel in (("C5 synth36", W.synth36_problem(), W.synth36_batch_x0(B), W.synth36_u_guess(40), 0, W.SYNTH_TARGET_VEL),
                                  ("C5q planar quad", W.planar_quad_problem(), W.planar_quad_batch_x0(B), W.planar_quad_u_guess(40), 0, W.QUAD_TARGET_VEL),
                                  ("C5q3d quad3d", W.quad3d_problem(), W.quad3d_batch_x0(B), W.quad3d_u_guess(40), 4, W.QUAD3D_TARGET_VEL)):
    n, N = q["Q"].shape[0], q["N"]
    step = np.zeros(n); step[col] = vel * q["dt"] * 4
    s = make_solver(q, B=B, jac="fd")
    s.SetInitialState(x0); s.SetInitialGuess(ug)
    s.Solve()
    it0, ls0 = s.iterations.copy(), s.ls_trials.copy()
    st = s.MPCRun(100, 4, target_step=step)
    cyc = s.stage_cycles.astype(float); it = s.iterations.astype(float); ls = s.ls_trials.astype(float) - ls0
    print(f"{name:18s} B {B}: MPC launch {s.stats.kernel_ms:7.2f} ms, iterations {int(it.sum())}, trials {int(ls.sum())} | cycles per iteration: line search "
          f"{(cyc[:, 0] / it).mean():8.0f} (per step {(cyc[:, 0] / ls).mean() / (N - 1):6.0f}) linearize {(cyc[:, 1] / it).mean():8.0f} backward {(cyc[:, 2] / it).mean():8.0f} "
          f"(per step {(cyc[:, 2] / it).mean() / (N - 1):6.0f}) all {(cyc[:, 3] / it).mean():8.0f}", flush=True)
