"""The slowest problems of the C2 batch, iteration by iteration: cycles of the forward pass (rollout + cost +
commit + linearization), the backward pass, and the whole iteration (in-kernel stopwatches, MI_F_ITER_CYCLES)."""
import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from drake_ddp_amd import workloads as W
from test_gpu_parity import make_solver
p = W.pendulum_problem()
s = make_solver(p, B=1024, jac="fd", hist_cap=16)
s.SetInitialState(W.pendulum_batch_x0(1024)); s.SetInitialGuess(np.zeros((1, p["N"] - 1)))
s.Solve(); s.Reset(); s.SetInitialGuess(np.zeros((1, p["N"] - 1))); s.Solve()
it = s.iterations; ic = s.iteration_cycles; cyc = s.stage_cycles; hist = s.history
print("kernel_ms", round(s.stats.kernel_ms, 4), "max iterations", it.max(), "problems at max", int((it == it.max()).sum()))
worst = np.argsort(-cyc[:, 3])[:3]
for b in worst:
    print("problem", b, "iterations", it[b], "total cycles", cyc[b, 3])
    for i in range(it[b]):
        fp, dv, bp, tot = ic[b, i]
        print("   iter %2d: forward %6d  derivs %5d  backward %5d  iteration %6d   eps %.3f ls %d" % (i + 1, fp, dv, bp, tot, hist[b, i, 1], hist[b, i, 2]))
    print("   sum of iterations", int(ic[b, :it[b], 3].sum()), " prologue+epilogue", int(cyc[b, 3] - ic[b, :it[b], 3].sum()))
