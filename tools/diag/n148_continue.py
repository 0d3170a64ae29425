"""Where does the device's on_indefinite="continue" solve of the planar quadruped at dt = 1.5e-3, N = 148 part from the reference?
First iteration on the device (rollout with zero gains, linearization), then the backward pass three ways on the SAME inputs -
device, NumPy oracle (np.linalg.inv), extended precision - and the eps = 1, 0.5, 0.25 rollouts (on the device) with each set of gains."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from drake_ddp_amd import workloads as W
from common import make_oracle, backward_extended
from test_gpu_parity import make_solver

N = int(os.environ.get("DIAG_N", "148"))
p = dict(W.planar_quad_problem(), dt=1.5e-3, N=N)
B = 2
x0, ug = W.planar_quad_batch_x0(4)[:B], W.planar_quad_u_guess(N)
s = make_solver(p, B=B, jac="fd", on_indefinite="continue")
s.SetInitialState(x0); s.SetInitialGuess(ug)
L0, eps0, tr0 = s.stage_forward(np.inf)
xb, ub, fx, fu = s.x_bar, s.u_bar, s.fx, s.fu
s.stage_backward()
Kd, kd, dVd, st = s.K, s.kappa, s.dV_coeff, s.status
print("first rollout cost", L0, "status after backward", st)
s2 = make_solver(p, B=B, jac="fd", on_indefinite="stop")
s2.SetInitialState(x0); s2.SetInitialGuess(ug)
s2.set_state(x_bar=xb, u_bar=ub, fx=fx, fu=fu)
s2.stage_backward()
Ks, ks_, st2 = s2.K, s2.kappa, s2.status
print("stop mode (unpivoted elimination throughout): status", st2, "max|kappa|", np.abs(ks_).max(axis=(1, 2)), "continue:", np.abs(kd).max(axis=(1, 2)))
for b in range(B):
    o = make_oracle(p, jacobian="fd")
    o.set_problem(x0[b], p["x_nom"], p["Q"], p["R"], p["Qf"], ug)
    o.x_bar, o.u_bar, o.fx, o.fu = xb[b].copy(), ub[b].copy(), fx[b].copy(), fu[b].copy()
    o.backward()
    Kx, kx, dx, cond = backward_extended(o)
    Kx, kx = np.asarray(Kx, float), np.asarray(kx, float)
    rel = lambda A, Bm, t: float(np.abs(A[..., t] - Bm[..., t]).max() / max(np.abs(Bm[..., t]).max(), 1e-300))
    ts = list(range(0, N - 1, max(1, (N - 1) // 10)))
    print(f"problem {b}: cond {cond:.1e}")
    print("  K  device vs ext :", ["%.1e" % rel(Kd[b], Kx, t) for t in ts])
    print("  K  numpy  vs ext :", ["%.1e" % rel(o.K, Kx, t) for t in ts])
    print("  kap device vs ext:", ["%.1e" % rel(kd[b], kx, t) for t in ts])
    print("  kap numpy  vs ext:", ["%.1e" % rel(o.kappa, kx, t) for t in ts])
    print("  max|K| device %.2e numpy %.2e ext %.2e ; max|kappa| device %.2e numpy %.2e ext %.2e" % (
        np.abs(Kd[b]).max(), np.abs(o.K).max(), np.abs(Kx).max(), np.abs(kd[b]).max(), np.abs(o.kappa).max(), np.abs(kx).max()))
    for t in range(0, 16):
        print("    t %2d |kappa| device %.3e stop-mode %.3e numpy %.3e ext %.3e   |K| device %.3e stop-mode %.3e numpy %.3e ext %.3e" % (
            t, np.abs(kd[b][:, t]).max(), np.abs(ks_[b][:, t]).max(), np.abs(o.kappa[:, t]).max(), np.abs(kx[:, t]).max(),
            np.abs(Kd[b][..., t]).max(), np.abs(Ks[b][..., t]).max(), np.abs(o.K[..., t]).max(), np.abs(Kx[..., t]).max()))
    print("  dV sum device %.6e numpy %.6e ext %.6e" % (dVd[b].sum(), o.dV.sum(), float(np.asarray(dx, float).sum())))
    if b == 0:
        gains = {"device": (Kd, kd), "numpy": (np.broadcast_to(o.K, Kd.shape).copy(), np.broadcast_to(o.kappa, kd.shape).copy()),
                 "extended": (np.broadcast_to(Kx, Kd.shape).copy(), np.broadcast_to(kx, kd.shape).copy())}
for name, (K_, k_) in gains.items():
    s.set_state(K=K_, kappa=k_, x_bar=xb, u_bar=ub)
    out = []
    for eps in (1.0, 0.5, 0.25, 0.0625):
        _, _, L, ex = s.stage_rollout(eps)
        out.append("%.4f" % L[0])
    print(f"rollouts of problem 0 with the {name} gains, eps = 1, .5, .25, .0625: {out}  (L_last {L0[0]:.4f})")
