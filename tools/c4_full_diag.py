"""Diagnostic behind tests/test_gpu_keypoints_quad3d_fullsize.py::test_c4_full_size_vs_c_oracle: C4 at its benchmarked size (B = 256, N = 200,
central differences) on the device against the C oracle, per problem: status, iterations, trials, the per-iteration
(eps, trials) history, and the final-cost deviation next to what a one-ulp change of x0 does to the oracle itself."""
import os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_parity import make_solver
from oracle import c_oracle, models_np as M
from drake_ddp_amd import workloads as W

c = W.cartpole_wall_problem()
B = 256
x0 = W.cartpole_wall_batch_x0(B)
ug = np.zeros((1, c["N"] - 1))
s = make_solver(c, B=B, jac="fd", hist_cap=64)
s.SetInitialState(x0); s.SetInitialGuess(ug)
x, u, _, L = s.Solve()
model = M.Model(c["model_id"], c["dt"])
r = c_oracle.solve_batch(model, c, x0, ug, hist_cap=64)
it_d, it_o = s.iterations, r["iters"]
print("status equal:", np.array_equal(s.status, r["status"]), "all converged:", (s.status == 0).all())
print("iterations equal:", int((it_d == it_o).sum()), "of", B, " trials equal:", int((s.ls_trials == r["ls"]).sum()))
h = s.history
lead = np.minimum(np.minimum(it_d, it_o), 64)
for k in (3, 5, 8, 12, 20, 64):
    ok = sum(np.array_equal(h[b, :min(k, lead[b]), 1:3], r["hist"][b, :min(k, lead[b]), 1:3]) for b in range(B))
    print(f"leading {k:2d} iterations (eps, trials) identical: {ok} of {B}")
rel = np.abs(L - r["cost"]) / np.abs(r["cost"])
print("final cost rel deviation: max %.2e median %.2e; where counts agree: max %.2e" % (rel.max(), np.median(rel), rel[it_d == it_o].max()))
# the oracle's own sensitivity: x0's pole angle moved by one ulp up / down
xp, xm = x0.copy(), x0.copy()
xp[:, 1] = np.nextafter(x0[:, 1], np.inf); xm[:, 1] = np.nextafter(x0[:, 1], -np.inf)
rp = c_oracle.solve_batch(model, c, xp, ug, hist_cap=64); rm = c_oracle.solve_batch(model, c, xm, ug, hist_cap=64)
env = np.maximum(np.abs(rp["cost"] - r["cost"]), np.abs(rm["cost"] - r["cost"])) / np.abs(r["cost"])
print("one-ulp envelope: max %.2e median %.2e;  oracle(+1ulp) iterations equal to oracle's: %d, (-1ulp): %d" %
      (env.max(), np.median(env), (rp["iters"] == it_o).sum(), (rm["iters"] == it_o).sum()))
ratio = rel / np.maximum(env, 1e-15)
print("device deviation / envelope: max %.1f, 90th pct %.1f, problems above 10x: %d" % (ratio.max(), np.percentile(ratio, 90), (ratio > 10).sum()))
bad = np.argsort(-ratio)[:8]
for b in bad:
    print(f"  b={b}: it dev/orc/+/- {it_d[b]}/{it_o[b]}/{rp['iters'][b]}/{rm['iters'][b]} rel {rel[b]:.2e} env {env[b]:.2e}")
