"""C2 (pendulum, B = 1024, cold solves): where a launch's cycles go, from the per-iteration stopwatches every problem logs
(MI_F_ITER_CYCLES: line search + commit + linearization | - | backward pass | iteration).  By iteration index, by trial count,
and along the slowest problems - the launch lasts as long as they do."""
import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from drake_ddp_amd import workloads as W
from test_gpu_parity import make_solver
p = W.pendulum_problem()
B = 1024
s = make_solver(p, B=B, jac="fd", hist_cap=16)
s.SetInitialState(W.pendulum_batch_x0(B)); s.SetInitialGuess(np.zeros((1, p["N"] - 1)))
for _ in range(3):                          # (cold-start solves, like the bench's steps)
    s.Reset(); s.SetInitialGuess(np.zeros((1, p["N"] - 1)))
    s.Solve()
it = s.iterations.astype(int); ic = s.iteration_cycles; h = s.history; tot = s.stage_cycles
print("kernel ms %.4f  iterations: mean %.2f max %d  histogram %s" % (s.stats.kernel_ms, it.mean(), it.max(), np.bincount(it).tolist()))
print("whole loop cycles: mean %.0f max %.0f (problem %d, %d iterations)" % (tot[:, 3].mean(), tot[:, 3].max(), tot[:, 3].argmax(), it[tot[:, 3].argmax()]))
for k in range(min(12, it.max())):
    live = it > k
    c = ic[live, k]; tr = h[live, k, 2]
    one = tr == 1
    print("iteration %2d: %4d problems | fwd (search+commit+linearize) mean %7.0f  bwd %6.0f  all %7.0f | first trial accepted: %4d at %7.0f cycles, backtracked: %4d at %7.0f (max trials %d)" % (
        k + 1, live.sum(), c[:, 0].mean(), c[:, 2].mean(), c[:, 3].mean(), one.sum(), c[one, 3].mean() if one.any() else 0, (~one).sum(), c[~one, 3].mean() if (~one).any() else 0, tr.max()))
slow = np.argsort(-tot[:, 3])[:6]
for b in slow:
    print("problem %4d: loop %7.0f cycles, %2d iterations; per iteration all: %s ; trials %s" % (b, tot[b, 3], it[b], np.round(ic[b, :it[b], 3]).astype(int).tolist(), h[b, :it[b], 2].astype(int).tolist()))
