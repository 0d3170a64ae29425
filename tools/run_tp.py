"""One throughput-mode (lane-per-problem) solve at the C2 shape, for rocprofv3 PMC passes: python tools/run_tp.py B"""
import sys, numpy as np
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from drake_ddp_amd import workloads as W
from drake_ddp_amd.ilqr import BatchedIterativeLQR
from drake_ddp_amd.models import ModelSystem
B = int(sys.argv[1]); mode = sys.argv[2] if len(sys.argv) > 2 else "throughput"
prob = W.pendulum_problem(); N = prob["N"]
s = BatchedIterativeLQR(ModelSystem(prob["model_id"], prob["dt"]), N, B, delta=prob["delta"], beta=prob["beta"], gamma=prob["gamma"], kernel_mode=mode, hist_cap=2)
s.SetTargetState(prob["x_nom"]); s.SetRunningCost(prob["Q"], prob["R"]); s.SetTerminalCost(prob["Qf"])
s.SetInitialState(W.pendulum_batch_x0(B)); s.SetInitialGuess(np.zeros((1, N - 1))); s._push_problem()
for _ in range(3):
    s.rearm(); st = s.solve_resident()
print(f"B={B} {mode}: kernel {st.kernel_ms:.3f} ms, iterations {st.total_iters}, algorithmic bytes {st.algorithmic_bytes:.4g}")
