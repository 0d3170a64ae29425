"""Per-iteration profile of the time-parallel rollout (needs a -DMI_PROF_NEWTON build): line-search cycles, Newton sweep
cycles and count, final-pass cycles.  python tools/newton_prof.py B [problem index to run alone]"""
import sys, numpy as np
sys.path.insert(0, ".")
from drake_ddp_amd import workloads as W
from drake_ddp_amd.ilqr import BatchedIterativeLQR
from drake_ddp_amd.models import ModelSystem
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
prob = W.pendulum_problem(); x0 = W.pendulum_batch_x0(B); N = prob["N"]
if len(sys.argv) > 2:                       # one problem of the batch, alone (the counters are per launch)
    x0 = x0[int(sys.argv[2]):int(sys.argv[2]) + 1]; B = 1
s = BatchedIterativeLQR(ModelSystem(prob["model_id"], prob["dt"]), N, B, delta=prob["delta"], beta=prob["beta"], gamma=prob["gamma"])
s.SetTargetState(prob["x_nom"]); s.SetRunningCost(prob["Q"], prob["R"]); s.SetTerminalCost(prob["Qf"])
s.SetInitialState(x0); s.SetInitialGuess(np.zeros((1, N - 1)))
s.Solve()
h = s.history; it = s.iterations
for b in sorted(set([0, B // 2, int(np.argmax(it))])):
    print("problem", b, "iters", it[b], "stage cycles", s.stage_cycles[b].tolist())
    ic = s.iteration_cycles
    for i in range(it[b]): print("   it %d: linesearch %6.0f  newton sweeps %6.0f (n=%d)  final pass %5.0f" % (i, *h[b, i]), " sweep updates", " ".join("%.1e" % v for v in ic[b, i]))
