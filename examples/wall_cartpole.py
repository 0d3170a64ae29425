#!/usr/bin/env python
"""Cart-pole that swings up against a wall (contact-rich), following the solver section of the
reference's cart_pole_with_wall.py (:142-166) — including its derivative key-point options
(:25-30).  The Drake diagram with hydroelastic contact becomes the device model
`CartPoleWithWall` (smooth penalty contact; see drake_ddp_amd/csrc/models.hpp)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from drake_ddp_amd import utils_derivs_interpolation  # noqa: E402
from drake_ddp_amd.ilqr import IterativeLinearQuadraticRegulator  # noqa: E402
from drake_ddp_amd.models import CartPoleWithWall  # noqa: E402

T, dt = 1.0, 1e-2
use_derivative_interpolation = "--interpolate" in sys.argv
keypoint_method, minN, maxN, jerk_threshold, iterative_error_threshold = "adaptiveJerk", 2, 10, 0.0007, 0.00005

x0 = np.array([0, np.pi + 0.5, 0.0, 0])
x_nom = np.array([0, np.pi, 0, 0])
Q = np.diag([0.1, 1, 0.01, 0.01])
R = 0.001 * np.eye(1)
Qf = np.diag([200, 200, 10, 10])

num_steps = int(T / dt)
interpolation_method = None
if use_derivative_interpolation:
    interpolation_method = utils_derivs_interpolation.derivs_interpolation(
        keypoint_method, minN, maxN, jerk_threshold, iterative_error_threshold)
ilqr = IterativeLinearQuadraticRegulator(CartPoleWithWall(dt), num_steps, beta=0.5,
                                         derivs_keypoint_method=interpolation_method, hist_cap=128)
ilqr.SetInitialState(x0)
ilqr.SetTargetState(x_nom)
ilqr.SetRunningCost(dt * Q, dt * R)
ilqr.SetTerminalCost(Qf)
ilqr.SetInitialGuess(np.zeros((1, num_steps - 1)))

states, inputs, solve_time, optimal_cost = ilqr.Solve()
print(f"Solved in {solve_time} seconds using iLQR")
print(f"Optimal cost: {optimal_cost}  (derivatives evaluated at {ilqr.percentage_derivs:.1f}% of the steps)")
