import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from drake_ddp_amd import workloads as W
from drake_ddp_amd.ilqr import BatchedIterativeLQR
from drake_ddp_amd.models import ModelSystem
q = W.synth36_problem(); N = q["N"]
s = BatchedIterativeLQR(ModelSystem(q["model_id"], q["dt"]), N, 1, delta=q["delta"], beta=q["beta"], gamma=q["gamma"], jacobian_mode="ad")
s.SetTargetState(q["x_nom"]); s.SetRunningCost(q["Q"], q["R"]); s.SetTerminalCost(q["Qf"])
s.SetInitialState(W.synth36_batch_x0(64)[:1]); s.SetInitialGuess(W.synth36_u_guess(N))
s.Solve()
H = s.history[0]; cap = H.shape[0]
w0 = H[cap - 4:cap].reshape(-1) / 39; w3 = H[cap - 8:cap - 4].reshape(-1) / 39
print("wave0 ticks/step:", {i: int(v) for i, v in enumerate(w0) if v})
print("wave3 ticks/step:", {i: int(v) for i, v in enumerate(w3) if v})
w1 = H[cap - 12:cap - 8].reshape(-1) / 39; w2 = H[cap - 16:cap - 12].reshape(-1) / 39
print("wave1 ticks/step:", {i: int(v) for i, v in enumerate(w1) if v})
print("wave2 ticks/step:", {i: int(v) for i, v in enumerate(w2) if v})
