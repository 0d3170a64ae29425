"""C4 (cart-pole with wall, beta = 0.5): line-search trial histogram and cycles per iteration."""
import sys, numpy as np
sys.path.insert(0, ".")
from drake_ddp_amd import workloads as W
from drake_ddp_amd.ilqr import BatchedIterativeLQR
from drake_ddp_amd.models import ModelSystem
B = 256
prob = W.cartpole_wall_problem(); x0 = W.cartpole_wall_batch_x0(B); N = prob["N"]
s = BatchedIterativeLQR(ModelSystem(prob["model_id"], prob["dt"]), N, B, delta=prob["delta"], beta=prob["beta"], gamma=prob["gamma"], hist_cap=64)
s.SetTargetState(prob["x_nom"]); s.SetRunningCost(prob["Q"], prob["R"]); s.SetTerminalCost(prob["Qf"])
s.SetInitialState(x0); s.SetInitialGuess(np.zeros((1, N - 1)))
s.Solve()
h = s.history; it = s.iterations; cyc = s.stage_cycles
tr = np.concatenate([h[b, :min(it[b], 64), 2].astype(int) for b in range(B)])
print("kernel ms", s.stats.kernel_ms, "iterations", it.sum(), "trials histogram:", {int(k): int((tr == k).sum()) for k in sorted(set(tr.tolist()))})
k = int(np.argmax(cyc[:, 3]))
print("critical problem", k, "iters", it[k], "cycles", cyc[k].tolist(), "trials/iter", h[k, :it[k], 2].astype(int).tolist())
print("mean cycles per iteration: ls %.0f lin %.0f bp %.0f" % tuple(cyc[:, i].sum() / it.sum() for i in range(3)))
