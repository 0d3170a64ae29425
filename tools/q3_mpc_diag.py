import os, sys
import numpy as np
ROOT = "/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_parity import make_solver
from oracle import c_oracle, models_np as M
from drake_ddp_amd import workloads as W
q = W.quad3d_problem(); B = 64
x0, ug = W.quad3d_batch_x0(B), W.quad3d_u_guess(q["N"])
step = np.zeros(37); step[4] = W.QUAD3D_TARGET_VEL * q["dt"] * 4
for jac in ("fd", "ad"):
    s = make_solver(q, B=B, jac=jac)
    s.SetInitialState(x0); s.SetInitialGuess(ug); s.Solve()
    st = s.MPCRun(100, 4, target_step=step)
    log = s.mpc_log
    if jac == "fd":
        r = c_oracle.mpc_batch(M.Model(q["model_id"], q["dt"]), q, x0, ug, 100, 4, target_step=step)
    same = log[:, :, -1] == r["log"][:, :, -1]
    first = np.array([np.argmin(same[b]) if not same[b].all() else 100 for b in range(B)])
    print(jac, "first differing re-solve per problem: min", first.min(), "median", np.median(first), "max", first.max(), " all-same problems:", (first == 100).sum())
    relL = np.abs(log[:, :, -2] - r["log"][:, :, -2]) / np.abs(r["log"][:, :, -2])
    for k in (1, 5, 10, 20, 40, 80, 99):
        print("   re-solve", k, "max rel cost err", relL[:, k].max(), "same iters", same[:, k].sum(), "max |dx0|", np.abs(log[:, k, :37] - r["log"][:, k, :37]).max())
    print("   iterations per re-solve (device) mean", log[:, :, -1].mean(), "max", log[:, :, -1].max(), "status", np.unique(s.status))
