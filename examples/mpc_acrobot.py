#!/usr/bin/env python
"""Receding-horizon acrobot control, batched over many initial states: the MPC loop of the
reference's acrobot.py (:131-162) on the device.  Shows the three ways to run it:
host loop (exactly the reference's code shape), on-device shift, and the single-launch loop."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from drake_ddp_amd import workloads as W  # noqa: E402
from drake_ddp_amd.ilqr import BatchedIterativeLQR  # noqa: E402
from drake_ddp_amd.models import Acrobot  # noqa: E402

B, num_resolves, replan_steps = 512, 50, 2
p = W.acrobot_problem()
num_steps = p["N"]


def make():
    s = BatchedIterativeLQR(Acrobot(p["dt"]), num_steps, B, beta=0.5)
    s.SetTargetState(p["x_nom"])
    s.SetRunningCost(p["Q"], p["R"])
    s.SetTerminalCost(p["Qf"])
    s.SetInitialState(W.acrobot_batch_x0(B))
    s.SetInitialGuess(np.zeros((1, num_steps - 1)))
    return s


# (1) the reference's loop shape: shift on the host, Solve()
s = make()
t0 = time.time()
x, u, _, _ = s.Solve()
for i in range(num_resolves):
    x0, u_guess = W.mpc_shift(x, u, replan_steps)
    s.SetInitialState(x0)
    s.SetInitialGuess(u_guess)
    x, u, _, cost = s.Solve()
t_host = time.time() - t0
ref_cost = cost.copy()

# (2) everything resident: one launch for the whole receding-horizon loop
s = make()
s.Solve()
t0 = time.time()
s.MPCRun(num_resolves, replan_steps)
t_dev = time.time() - t0
assert np.allclose(s.cost, ref_cost, rtol=1e-9)
print(f"{B} acrobots x {num_resolves} re-solves: host loop {t_host*1e3:.1f} ms, device loop {t_dev*1e3:.1f} ms, "
      f"best cost {s.cost.min():.4f}")
