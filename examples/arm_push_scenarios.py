#!/usr/bin/env python
"""The scenario switch of the reference's panda_fr3.py (:20-58: the scripts' start pose, ball target and initial guess change,
the solver section :201-228 does not) on the build's arm + ball model: "side" (the ball 15 cm along +y, kinova_gen3.py's
scenario too) and "forward" (20 cm along +x, the hand starting behind the ball) - one cold Solve() each like the script, then the
receding-horizon loop on the device for a batch of perturbed starts, which is what carries the ball the rest of the way
when one 0.5 s horizon is not enough.  "lift" needs the reference's whole-arm hydroelastic wrap around the ball; the build's
single point contact cannot carry it (drake_ddp_amd/workloads.py: arm27_scenario)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from drake_ddp_amd import workloads as W  # noqa: E402
from drake_ddp_amd.ilqr import BatchedIterativeLQR, IterativeLinearQuadraticRegulator  # noqa: E402
from drake_ddp_amd.models import ArmAndBall  # noqa: E402

T, dt = 0.5, 1e-2
num_steps = int(T / dt)
system_ = ArmAndBall(dt)
for scenario in (sys.argv[1:] or ["side", "forward"]):
    p, x0, u_guess = W.arm27_scenario(scenario, num_steps)
    axis = 12 if scenario == "side" else 11
    # ---- panda_fr3.py:201-228
    ilqr = IterativeLinearQuadraticRegulator(system_, num_steps, beta=0.5, delta=1e-3, gamma=0, verbose=False)
    ilqr.SetInitialState(x0)
    ilqr.SetTargetState(p["x_nom"])
    ilqr.SetRunningCost(p["Q"], p["R"])
    ilqr.SetTerminalCost(p["Qf"])
    ilqr.SetInitialGuess(u_guess)
    states, inputs, solve_time, optimal_cost = ilqr.Solve()
    print(f"[{scenario}] Solved in {solve_time} seconds using iLQR; optimal cost {optimal_cost:.4f}; "
          f"ball {'xyz'[axis - 11]} {states[axis, 0]:.3f} -> {states[axis, -1]:.3f} m (target {p['x_nom'][axis]:.3f})")
    # ---- the same scenario, 32 perturbed starts, re-planned every 5 steps for another 1.5 s
    B, num_resolves, replan_steps = 32, 30, 5
    rng = np.random.default_rng(1)
    xb = np.tile(x0, (B, 1))
    xb[:, 11:13] += rng.uniform(-0.005, 0.005, (B, 2))
    batch = BatchedIterativeLQR(system_, num_steps, B, beta=0.5, delta=1e-3, gamma=0)
    batch.SetTargetState(p["x_nom"])
    batch.SetRunningCost(p["Q"], p["R"])
    batch.SetTerminalCost(p["Qf"])
    batch.SetInitialState(xb)
    batch.SetInitialGuess(u_guess)
    batch.Solve()
    stats = batch.MPCRun(num_resolves, replan_steps)
    log = batch.mpc_log
    print(f"[{scenario}] {B} starts, {num_resolves} re-plans: {stats.total_iters} iLQR iterations in the re-solves, all converged: {stats.n_converged == B}; "
          f"ball {'xyz'[axis - 11]} at the last re-plan's start {log[:, -1, axis].min():.3f}..{log[:, -1, axis].max():.3f} m")
