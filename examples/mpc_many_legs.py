#!/usr/bin/env python
"""Receding-horizon control of the cheetah-SHAPED model (n=36, m=12) with a moving target,
following the loop of the reference's mini_cheetah.py (:147-201): every re-solve shifts the
control tape by `replan_steps`, restarts from x[:, replan_steps] and advances the target by
target_vel*dt*replan_steps.  Runs a batch of seeds on the workgroup-per-problem kernels."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from drake_ddp_amd import workloads as W  # noqa: E402
from drake_ddp_amd.ilqr import BatchedIterativeLQR  # noqa: E402
from drake_ddp_amd.models import Synth36  # noqa: E402

B, num_resolves, replan_steps = 64, 20, 4
p = W.synth36_problem()
num_steps, dt = p["N"], p["dt"]

ilqr = BatchedIterativeLQR(Synth36(dt), num_steps, B, beta=0.5, delta=1e-2, gamma=0)
x_nom = p["x_nom"].copy()
ilqr.SetTargetState(x_nom)
ilqr.SetRunningCost(p["Q"], p["R"])
ilqr.SetTerminalCost(p["Qf"])
ilqr.SetInitialState(W.synth36_batch_x0(B))
ilqr.SetInitialGuess(W.synth36_u_guess(num_steps))

st = time.time()
x, u, _, cost = ilqr.Solve()
step = np.zeros(36)
step[0] = W.SYNTH_TARGET_VEL * dt * replan_steps           # move the 'base x position' target
stats = ilqr.MPCRun(num_resolves, replan_steps, target_step=step)
print(f"{B} seeds x (1 + {num_resolves}) solves in {(time.time() - st) * 1e3:.1f} ms; "
      f"{stats.total_iters} iLQR iterations in the re-solves; final cost range "
      f"[{ilqr.cost.min():.4f}, {ilqr.cost.max():.4f}]")
