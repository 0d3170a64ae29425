import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from drake_ddp_amd import workloads as W
from test_gpu_parity import make_solver
B = 2
q, x0, ug = W.planar_quad_problem(), W.planar_quad_batch_x0(4)[:B], W.planar_quad_u_guess(40)
# reference path: cold solve, shift on the device, ONE forward pass (MODE_FORWARD)
s = make_solver(q, B=B, jac="fd")
s.SetInitialState(x0); s.SetInitialGuess(ug); s.Solve()
xb0, ub0, K0 = s.x_bar.copy(), s.u_bar.copy(), s.K.copy()
s.MPCShift(4)
import ctypes as C
from drake_ddp_amd import _capi
Linf = np.full(B, np.inf)
_capi.check(s._lib.mi_ilqr_forward(s._h, _capi.ptr(Linf)), "forward")
xr, ur, Lr = s.x_bar.copy(), s.u_bar.copy(), s.history[:, 0, 0].copy()
for rep in range(3):
    t = make_solver(q, B=B, jac="fd", max_iters=1)
    t.SetInitialState(x0); t.SetInitialGuess(ug)
    t2 = make_solver(q, B=B, jac="fd"); t2.SetInitialState(x0); t2.SetInitialGuess(ug); t2.Solve()
    # give t the converged state of the cold solve, then one in-kernel re-solve capped at one iteration
    t.Solve()   # (max_iters=1: leaves a one-iteration state) -> overwrite with the converged one
    t.set_state(x_bar=xb0, u_bar=ub0, K=K0, kappa=t2.kappa, dV_coeff=t2.dV_coeff, fx=t2.fx, fu=t2.fu)
    t.SetInitialState(x0); t._push_problem()
    st = t.MPCRun(1, 4)
    xm, um = t.x_bar, t.u_bar
    du = np.abs(um - ur).max(axis=1); dx = np.abs(xm - xr).max(axis=1)
    print(rep, "status", t.status.tolist(), "first-iteration cost MPC kernel", t.history[:, 0, 0].round(5).tolist(), "forward stage", Lr.round(5).tolist())
    for b in range(B):
        bad_u = np.nonzero(du[b] > 1e-9)[0]; bad_x = np.nonzero(dx[b] > 1e-9)[0]
        print("   problem", b, "first u step off:", bad_u[:5], "max", du[b].max(), "| first x step off:", bad_x[:5], "max", dx[b].max())
