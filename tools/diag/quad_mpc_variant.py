import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from drake_ddp_amd import workloads as W
from test_gpu_parity import make_solver
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
q, x0, ug = W.planar_quad_problem(), W.planar_quad_batch_x0(B), W.planar_quad_u_guess(40)
step = np.zeros(36); step[0] = W.QUAD_TARGET_VEL * q["dt"] * 4
s = make_solver(q, B=B, jac="fd")
s.SetInitialState(x0); s.SetInitialGuess(ug)
s.Solve()
print("cold solve: iterations", s.iterations[:8], "status", np.unique(s.status, return_counts=True), "cost", s.cost[:4])
st = s.MPCRun(100, 4, target_step=step)
print("MPC: kernel ms %.2f total iters %d status" % (st.kernel_ms, st.total_iters), np.unique(s.status, return_counts=True), "n_internal", st.n_internal, "n_not_pd", st.n_not_pd, "ls_failed", st.n_ls_failed)
lg = s.mpc_log
print("re-solves logged per problem (nonzero rows):", (lg[:, :, -1] > 0).sum(axis=1)[:16])
for label, kw in (("no target step", dict(target_step=None)), ("zero step", dict(target_step=np.zeros(36))), ("step", dict(target_step=step))):
    s = make_solver(q, B=4, jac="fd")
    s.SetInitialState(x0[:4]); s.SetInitialGuess(ug)
    s.Solve()
    st = s.MPCRun(3, 4, **kw)
    lg = s.mpc_log
    print(label, "status", s.status, "iters", s.iterations, "log cost", lg[:, :, -2].round(4).tolist(), "x0[0] of re-solve 0:", lg[0, 0, :3])
