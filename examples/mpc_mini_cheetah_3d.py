#!/usr/bin/env python
"""The solver section of the reference's mini_cheetah.py (:163-213) on the build's 3-D quadruped: the same state layout
(unit quaternion | base position | 12 joints | 18 velocities: n = 37, m = 12), cost weights, standing-torque initial
guess, forward-velocity target that advances by target_vel * dt * replan_steps with every re-solve, and receding-horizon
loop - here for a batch of perturbed stances, cold solve + all re-solves in one launch on the workgroup-per-problem
(matrix-core) kernels.  Drake's plant is replaced by drake_ddp_amd.models.Quadruped3D (Drake cannot run on the GPU)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from drake_ddp_amd import workloads as W  # noqa: E402
from drake_ddp_amd.ilqr import BatchedIterativeLQR  # noqa: E402
from drake_ddp_amd.models import Quadruped3D  # noqa: E402

B, num_resolves, replan_steps = 16, 25, 4
p = W.quad3d_problem()
num_steps, dt = p["N"], p["dt"]

ilqr = BatchedIterativeLQR(Quadruped3D(dt), num_steps, B, beta=0.5, delta=1e-2, gamma=0)
ilqr.SetTargetState(p["x_nom"])
ilqr.SetRunningCost(p["Q"], p["R"])
ilqr.SetTerminalCost(p["Qf"])
x0 = W.quad3d_batch_x0(B)
ilqr.SetInitialState(x0)
ilqr.SetInitialGuess(W.quad3d_u_guess(num_steps))

st = time.time()
x, u, _, cost = ilqr.Solve()
step = np.zeros(37)
step[4] = W.QUAD3D_TARGET_VEL * dt * replan_steps          # x_nom[4] += target_vel * delta_t (mini_cheetah.py:151-156)
stats = ilqr.MPCRun(num_resolves, replan_steps, target_step=step)
log = ilqr.mpc_log
travelled = log[:, -1, 4] - x0[:, 4]
qn = np.linalg.norm(log[:, -1, 0:4], axis=1)
print(f"{B} stances x (1 + {num_resolves}) solves in {(time.time() - st) * 1e3:.1f} ms; "
      f"{stats.total_iters} iLQR iterations in the re-solves, all converged: {stats.n_converged == B}; "
      f"trunk moved {travelled.min():.3f}..{travelled.max():.3f} m in {num_resolves * replan_steps * dt:.2f} s, "
      f"height {log[:, -1, 6].min():.3f}..{log[:, -1, 6].max():.3f} m, |quaternion| {qn.min():.4f}..{qn.max():.4f}")
