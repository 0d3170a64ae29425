#!/usr/bin/env python
"""The solver section of the reference's kinova_gen3.py (:254-284; panda_fr3.py:201-228 is the same) on the build's arm + ball
model: the same state layout (7 joint angles | the ball's quaternion and position | 13 velocities: n = 27, m = 7), horizon
(T = 0.5 s, dt = 1e-2), cost ("side" scenario: move the ball 15 cm along +y), delta = 1e-3, beta = 0.5, gravity-compensation
initial guess, Solve() and SaveSolution() - then the same problem as a BATCH of perturbed starts with a receding-horizon
loop on the device.  Drake's plant is replaced by drake_ddp_amd.models.ArmAndBall (Drake cannot run on the GPU); the
kernels are the mid-size workgroup-per-problem family (n <= 32, any m <= 16).

    python examples/arm_reach.py [--coupled]     --coupled: the arm with coupled joint dynamics (ArmAndBallCoupled, MI_MODEL_ARM27C:
                                                 dense mass matrix, velocity-product terms) instead of per-joint inertias"""
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from drake_ddp_amd import workloads as W  # noqa: E402
from drake_ddp_amd.ilqr import BatchedIterativeLQR, IterativeLinearQuadraticRegulator  # noqa: E402
from drake_ddp_amd.models import ArmAndBall, ArmAndBallCoupled  # noqa: E402

coupled = "--coupled" in sys.argv
p = W.arm27c_problem() if coupled else W.arm27_problem()
num_steps, dt = p["N"], p["dt"]
system_ = (ArmAndBallCoupled if coupled else ArmAndBall)(dt)
u_guess = (W.arm27c_u_guess if coupled else W.arm27_u_guess)(num_steps)

# ---- kinova_gen3.py:254-284
ilqr = IterativeLinearQuadraticRegulator(system_, num_steps, beta=0.5, delta=1e-3, gamma=0, derivs_keypoint_method=None)
ilqr.SetInitialState(W.arm27_start())
ilqr.SetTargetState(p["x_nom"])
ilqr.SetRunningCost(p["Q"], p["R"])
ilqr.SetTerminalCost(p["Qf"])
ilqr.SetInitialGuess(u_guess)                           # gravity compensation (+ a push on the base joint: workloads.py)
states, inputs, solve_time, optimal_cost = ilqr.Solve()
print(f"Solved in {solve_time} seconds using iLQR")
print(f"Optimal cost: {optimal_cost}")
print(f"ball: y {states[12, 0]:.3f} -> {states[12, -1]:.3f} m (target {p['x_nom'][12]:.3f}), height {states[13, -1]:.3f} m")
save_file = os.path.join(tempfile.gettempdir(), "side.npz")
ilqr.SaveSolution(save_file)
data = np.load(save_file)
print("saved", save_file, {k: data[k].shape for k in data.files})

# ---- the same task for a batch of perturbed starts, re-planned every 5 steps
B, num_resolves, replan_steps = 64, 10, 5
batch = BatchedIterativeLQR(system_, num_steps, B, beta=0.5, delta=1e-3, gamma=0)
batch.SetTargetState(p["x_nom"])
batch.SetRunningCost(p["Q"], p["R"])
batch.SetTerminalCost(p["Qf"])
batch.SetInitialState(W.arm27_batch_x0(B))
batch.SetInitialGuess(u_guess)
st = time.time()
x, u, _, cost = batch.Solve()
it0 = int(batch.iterations.sum())
stats = batch.MPCRun(num_resolves, replan_steps)
log = batch.mpc_log
print(f"{B} starts x (1 + {num_resolves}) solves in {(time.time() - st) * 1e3:.1f} ms: {it0} + {stats.total_iters} iLQR iterations, "
      f"all converged: {stats.n_converged == B}; ball y after the cold solve's plan {x[:, 12, -1].min():.3f}..{x[:, 12, -1].max():.3f} m, "
      f"at the last re-plan's start {log[:, -1, 12].min():.3f}..{log[:, -1, 12].max():.3f} m")
