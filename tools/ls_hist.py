"""Line-search trial counts per iteration at the C2 shape: which problems backtrack, and how far."""
import sys, numpy as np
sys.path.insert(0, ".")
from drake_ddp_amd import workloads as W
from drake_ddp_amd.ilqr import BatchedIterativeLQR
from drake_ddp_amd.models import ModelSystem
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
prob = W.pendulum_problem(); x0 = W.pendulum_batch_x0(B); N = prob["N"]
s = BatchedIterativeLQR(ModelSystem(prob["model_id"], prob["dt"]), N, B, delta=prob["delta"], beta=prob["beta"], gamma=prob["gamma"])
s.SetTargetState(prob["x_nom"]); s.SetRunningCost(prob["Q"], prob["R"]); s.SetTerminalCost(prob["Qf"])
s.SetInitialState(x0); s.SetInitialGuess(np.zeros((1, N - 1)))
s.Solve()
h = s.history; it = s.iterations; cyc = s.stage_cycles
tr = [h[b, :it[b], 2].astype(int) for b in range(B)]
allt = np.concatenate(tr)
print("iterations:", len(allt), " trials histogram:", {int(k): int((allt == k).sum()) for k in sorted(set(allt.tolist()))})
order = np.argsort(-cyc[:, 3])[:8]
for b in order:
    print(f"problem {b}: cycles {cyc[b,3]} iters {it[b]} trials/iter {tr[b].tolist()}")
