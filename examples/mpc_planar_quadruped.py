#!/usr/bin/env python
"""Receding-horizon control of the planar quadruped (trunk + four 3-link legs + passive tail, ground contact;
n = 36, m = 12) with the loop of the reference's mini_cheetah.py (:147-201): standing-torque initial guess
(`u_stand`), a forward-velocity target that advances with every re-solve, `replan_steps` controls dropped per
re-solve.  A batch of perturbed stances on the workgroup-per-problem kernels; with this few problems per GPU a
cluster of workgroups shares each problem's linearization.  The model can declare a step infeasible (a joint
velocity beyond `v_max`): such a line-search trial costs +inf, as when Drake's update throws (ilqr.py:315-323)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from drake_ddp_amd import workloads as W  # noqa: E402
from drake_ddp_amd.ilqr import BatchedIterativeLQR  # noqa: E402
from drake_ddp_amd.models import PlanarQuadruped  # noqa: E402

B, num_resolves, replan_steps = 16, 40, 4
p = W.planar_quad_problem()
num_steps, dt = p["N"], p["dt"]

ilqr = BatchedIterativeLQR(PlanarQuadruped(dt), num_steps, B, beta=0.5, delta=1e-2, gamma=0)
ilqr.SetTargetState(p["x_nom"])
ilqr.SetRunningCost(p["Q"], p["R"])
ilqr.SetTerminalCost(p["Qf"])
x0 = W.planar_quad_batch_x0(B)
ilqr.SetInitialState(x0)
ilqr.SetInitialGuess(W.planar_quad_u_guess(num_steps))

st = time.time()
x, u, _, cost = ilqr.Solve()
step = np.zeros(36)
step[0] = W.QUAD_TARGET_VEL * dt * replan_steps            # the trunk's x target moves on
stats = ilqr.MPCRun(num_resolves, replan_steps, target_step=step)
log = ilqr.mpc_log
travelled = log[:, -1, 0] - x0[:, 0]
print(f"{B} stances x (1 + {num_resolves}) solves in {(time.time() - st) * 1e3:.1f} ms; "
      f"{stats.total_iters} iLQR iterations in the re-solves, all converged: {stats.n_converged == B}; "
      f"trunk moved {travelled.min():.3f}..{travelled.max():.3f} m in {num_resolves * replan_steps * dt:.2f} s, "
      f"height {log[:, -1, 1].min():.3f}..{log[:, -1, 1].max():.3f} m")
