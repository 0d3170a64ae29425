import sys, numpy as np
sys.path.insert(0, ".")
from drake_ddp_amd import workloads as W
from drake_ddp_amd.ilqr import BatchedIterativeLQR
from drake_ddp_amd.models import ModelSystem
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
prob = W.pendulum_problem(); x0 = W.pendulum_batch_x0(B); N = prob["N"]
s = BatchedIterativeLQR(ModelSystem(prob["model_id"], prob["dt"]), N, B, delta=prob["delta"], beta=prob["beta"], gamma=prob["gamma"])
s.SetTargetState(prob["x_nom"]); s.SetRunningCost(prob["Q"], prob["R"]); s.SetTerminalCost(prob["Qf"])
s.SetInitialState(x0); s.SetInitialGuess(np.zeros((1, N - 1)))
s.Solve()
h = s.history; it = s.iterations
for b in sorted(set([0, B // 2, int(np.argmax(it))])):
    print("problem", b, "iters", it[b], "stage cycles", s.stage_cycles[b].tolist())
    for i in range(it[b]): print("   it %d: linesearch %6.0f  newton sweeps %6.0f (n=%d)  final pass %5.0f" % (i, *h[b, i]))
