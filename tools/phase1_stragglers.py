"""Which problems of C2 define the duration of the first phase of a two-phase solve: cycles of every problem's first k iterations."""
import os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_parity import make_solver
from drake_ddp_amd import workloads as W
p = W.pendulum_problem(); B = 1024
s = make_solver(p, B=B, jac="fd", hist_cap=16)
s.SetInitialState(W.pendulum_batch_x0(B)); s.SetInitialGuess(np.zeros((1, p["N"] - 1)))
s.Solve(); s.rearm(cold=True); s.solve_resident()
ic = s.iteration_cycles[:, :, 3]; h = s.history; it = s.iterations
for k in (4, 5, 6, 7, 8):
    first = np.array([ic[b, :min(k, it[b])].sum() for b in range(B)])
    rem = (it > k).sum()
    top = np.argsort(-first)[:5]
    print(f"k={k}: unfinished after k: {rem}; first-k cycles: median {np.median(first):.0f} p99 {np.percentile(first,99):.0f} max {first.max():.0f};",
          "top:", [(int(b), int(first[b]), int(it[b]), h[b, :min(k, it[b]), 2].astype(int).tolist()) for b in top[:3]])
bt = [(int(b), int(i), int(h[b, i, 2]), int(ic[b, i])) for b in range(B) for i in range(min(it[b], 16)) if h[b, i, 2] > 1]
print("backtracking iterations (problem, iteration index, trials, cycles):", bt)
tot = np.array([ic[b, :min(it[b], 16)].sum() for b in range(B)])
print("whole-solve cycles: max", tot.max(), "of problems with a backtracking iteration:", [int(tot[b]) for b, _, _, _ in bt])
