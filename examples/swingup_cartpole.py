#!/usr/bin/env python
"""Cart-pole swing-up from hanging at rest: the solver section of the reference's cart_pole.py (:95-125) with its
parameters (:21-47: T = 2 s, dt = 0.01, beta = 0.9, Q = diag(10, 10, 0.1, 0.1), R = 0.001, Qf = diag(100, 100, 10, 10)) and
its derivative-interpolation options (--interpolate: adaptiveJerk, minN = 5, maxN = 10, jerk threshold 1e-4).  Drake's
cart_pole.sdf plant becomes the device model `CartPole` (drake_ddp_amd/csrc/models.hpp)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from drake_ddp_amd import utils_derivs_interpolation  # noqa: E402
from drake_ddp_amd.ilqr import IterativeLinearQuadraticRegulator  # noqa: E402
from drake_ddp_amd.models import CartPole  # noqa: E402

T, dt = 2.0, 0.01
use_derivative_interpolation = "--interpolate" in sys.argv
keypoint_method, minN, maxN, jerk_threshold, iterative_error_threshold = "adaptiveJerk", 5, 10, 1e-4, 0.00005

x0 = np.array([0, 0, 0, 0])
x_nom = np.array([0, np.pi, 0, 0])
Q = np.diag([10, 10, 0.1, 0.1])
R = 0.001 * np.eye(1)
Qf = np.diag([100, 100, 10, 10])

num_steps = int(T / dt)
interpolation_method = None
if use_derivative_interpolation:
    interpolation_method = utils_derivs_interpolation.derivs_interpolation(
        keypoint_method, minN, maxN, jerk_threshold, iterative_error_threshold)
ilqr = IterativeLinearQuadraticRegulator(CartPole(dt), num_steps, beta=0.9, derivs_keypoint_method=interpolation_method,
                                         hist_cap=512)
ilqr.SetInitialState(x0)
ilqr.SetTargetState(x_nom)
ilqr.SetRunningCost(dt * Q, dt * R)
ilqr.SetTerminalCost(Qf)
ilqr.SetInitialGuess(np.zeros((1, num_steps - 1)))

states, inputs, solve_time, optimal_cost = ilqr.Solve()
print(f"Solved in {solve_time} seconds using iLQR")
print(f"Optimal cost: {optimal_cost}  (derivatives evaluated at {ilqr.percentage_derivs:.1f}% of the steps)")
print(f"pole angle {states[1, 0]:.3f} -> {states[1, -1]:.3f} rad (target {np.pi:.3f}), cart at {states[0, -1]:.3f} m, "
      f"peak force {np.abs(inputs).max():.1f} N")
