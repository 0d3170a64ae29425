"""A/B of the two-phase n = 2 solve (ilqr_wide.hpp) on C2: python tools/two_phase_ab.py   (MI_ILQR_PHASE_CAP=k: 0 = single phase)
Prints iterations/s of pipelined cold-start solves and checks all 1024 problems against the C oracle."""
import os, sys, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_parity import make_solver
from oracle import c_oracle, models_np as M
from drake_ddp_amd import workloads as W
p = W.pendulum_problem(); B = 1024
x0 = W.pendulum_batch_x0(B)
s = make_solver(p, B=B, jac="fd", hist_cap=16)
s.SetInitialState(x0); s.SetInitialGuess(np.zeros((1, p["N"] - 1)))
x, u, _, L = s.Solve()
r = c_oracle.solve_batch(M.Model(p["model_id"], p["dt"]), p, x0, np.zeros((1, p["N"] - 1)))
same = (s.iterations == r["iters"]) & (s.ls_trials == r["ls"]) & (s.status == r["status"])
rel = np.abs(L - r["cost"]) / np.abs(r["cost"])
print("cap", os.environ.get("MI_ILQR_PHASE_CAP", "default"), "identical counts", int(same.sum()), "of", B, "max rel cost", rel.max(), "max |dx|", np.abs(x - r["x_bar"]).max(),
      "max |dK| rel", np.abs(s.K - r["K"]).max() / np.abs(r["K"]).max(), "status", np.unique(s.status), "iters max", s.iterations.max())
s.set_timing(4)
for rep in range(6):
    t0 = time.perf_counter()
    for _ in range(32):
        s.rearm(cold=True); s.solve_resident_async()
    st = s.collect(32)
    dt = time.perf_counter() - t0
    it = sum(q.total_iters for q in st)
    km = [q.kernel_ms for q in st if q.kernel_ms > 0]
print("  %.2f M it/s, %.4f ms/step, kernel(s) %.4f ms" % (it / dt / 1e6, 1e3 * dt / 32, sum(km) / len(km)))
c = s.stage_cycles; it = s.iterations
slow = np.argsort(-it)[:3]
print("  slowest problems: iters", it[slow], "cycles (ls, lin, bp, total)", c[slow].tolist())
ic = s.iteration_cycles[slow[0]][:int(min(it[slow[0]], 16))]
print("  per-iteration cycles of the slowest:", ic[:, 3].astype(int).tolist())
