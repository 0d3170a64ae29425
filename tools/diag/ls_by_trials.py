"""Line-search cycles of an iteration against its reference trial count (arm + ball, cold solves, B = 64): what a search of k trials costs
under the policy / candidate-group switches of the environment."""
import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from drake_ddp_amd import workloads as W
from test_gpu_parity import make_solver
p = W.arm27_problem(); B = 64
s = make_solver(p, B=B, jac="fd"); s.SetInitialState(W.arm27_batch_x0(B)); s.SetInitialGuess(W.arm27_u_guess(p["N"]))
s.Solve()
it = s.iterations
ic = s.iteration_cycles          # (B, hist_cap, 4)
h = s.history if hasattr(s, "history") else None
rows = {}
for b in range(B):
    for k in range(min(int(it[b]), ic.shape[1])):
        tr = int(round(h[b, k, 2])) if h is not None else -1
        rows.setdefault(tr, []).append(ic[b, k, 0])
print("policy", os.environ.get("MI_ILQR_SPEC", "1"), "groups", os.environ.get("MI_ILQR_LS_GROUPS", "1"), "iterations", int(it.sum()), "kernel_ms %.3f" % s.stats.kernel_ms)
for tr in sorted(rows):
    v = np.array(rows[tr]); print("  trials %2d: %4d searches, line search %8.0f cycles (min %8.0f max %8.0f)" % (tr, len(v), v.mean(), v.min(), v.max()))
