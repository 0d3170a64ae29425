#!/usr/bin/env python
"""Pendulum swing-up with the drop-in solver — the solver section of the reference's
pendulum.py (:84-103) with only the import and the `system` argument changed: the Drake
plant becomes a device model descriptor.  No visualizer (Drake is not needed)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from drake_ddp_amd.ilqr import IterativeLinearQuadraticRegulator  # noqa: E402
from drake_ddp_amd.models import Pendulum  # noqa: E402

T, dt = 2.0, 1e-2
x0 = np.array([0, 0])
x_nom = np.array([np.pi, 0])
Q = 0.01 * np.diag([0, 1])
R = 0.01 * np.eye(1)
Qf = 100 * np.diag([1, 1])

num_steps = int(T / dt)
ilqr = IterativeLinearQuadraticRegulator(Pendulum(dt), num_steps)
ilqr.SetInitialState(x0)
ilqr.SetTargetState(x_nom)
ilqr.SetRunningCost(dt * Q, dt * R)
ilqr.SetTerminalCost(Qf)
ilqr.SetInitialGuess(np.zeros((1, num_steps - 1)))

states, inputs, solve_time, optimal_cost = ilqr.Solve()
print(f"Solved in {solve_time} seconds using iLQR")
print(f"Optimal cost: {optimal_cost}")
print(f"final state: {states[:, -1]}  (target {x_nom})")
