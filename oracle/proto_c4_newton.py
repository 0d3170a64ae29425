#!/usr/bin/env python
"""ORACLE-side experiment (test infrastructure, not product): would a TIME-PARALLEL rollout pay for C4 (cart-pole with wall,
n = 4, N = 200, beta = 0.5)?  The n = 2 kernels roll a line-search trial out as Newton's method on the whole trajectory
(drake_ddp_amd/csrc/ilqr_small.hpp: rollout_newton); this script counts, with NumPy, how many Newton sweeps the same scheme
needs on C4's stiff contact (k / sigma^2 = 2e7) for the candidates the reference's line search actually visits
(ilqr.py:300-337: eps = 1, 0.5, 0.25, ...), to an update below 1e-10.

A sweep linearizes the closed-loop step g_t(x) = f(x, u_bar_t - eps kappa_t - K_t (x - x_bar_t)) around the current guess
and solves the linear recurrence d_{t+1} = G_t d_t + r_t exactly (what the device's prefix scan of affine maps computes).
Starting guess: the nominal trajectory, or the backward pass's own linear prediction (the device's predictor).

    PYTHONDONTWRITEBYTECODE=1 python oracle/proto_c4_newton.py [n_problems]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import models_np as M, problems as P  # noqa: E402
from oracle.ilqr_np import OracleILQR  # noqa: E402


def newton_rollout(o, eps, predictor, tol=1e-10, max_sweeps=40):
    """Sweeps until the trajectory moves by less than tol; returns (sweeps or -1, max defect against the exact rollout)."""
    n, m, N = o.n, o.m, o.N
    model = o.model
    X = o.x_bar.copy()
    X[:, 0] = o.x0
    if predictor:                                  # dx_{t+1} = (fx - fu K) dx_t - eps fu kappa around the nominal trajectory
        d = o.x0 - o.x_bar[:, 0]
        for t in range(N - 1):
            X[:, t] = o.x_bar[:, t] + d
            d = (o.fx[:, :, t] - o.fu[:, :, t] @ o.K[:, :, t]) @ d - eps * o.fu[:, :, t] @ o.kappa[:, t]
        X[:, N - 1] = o.x_bar[:, N - 1] + d
    for sweep in range(1, max_sweeps + 1):
        G = np.zeros((N - 1, n, n))
        r = np.zeros((N - 1, n))
        for t in range(N - 1):
            u = o.u_bar[:, t] - eps * o.kappa[:, t] - o.K[:, :, t] @ (X[:, t] - o.x_bar[:, t])
            fx, fu = model.jac_ad(X[:, t], u)
            G[t] = fx - fu @ o.K[:, :, t]
            r[t] = model.step_unchecked(X[:, t], u) - X[:, t + 1]
        d = np.zeros(n)
        upd = 0.0
        for t in range(N - 1):
            X[:, t] = X[:, t] + d
            upd = max(upd, float(np.max(np.abs(d))))
            d = G[t] @ d + r[t]
        X[:, N - 1] = X[:, N - 1] + d
        upd = max(upd, float(np.max(np.abs(d))))
        if not np.isfinite(upd):
            return -1, np.inf
        if upd < tol:
            xs, _, _, _ = o.rollout(eps)
            return sweep, float(np.max(np.abs(xs - X)))
    return -1, np.inf


def main():
    nprob = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    prob = P.cartpole_wall_problem(N=200)
    x0s = P.cartpole_wall_batch_x0(256)[:nprob]
    model = M.Model(prob["model_id"], prob["dt"])
    stats = {True: [], False: []}           # per backtracking iteration: list of sweeps per visited candidate
    fails = {True: 0, False: 0}
    n_iter = n_bt = 0
    trials_hist = []
    for b in range(nprob):
        o = OracleILQR(model, prob["N"], prob["delta"], prob["beta"], prob["gamma"], jacobian="ad")
        o.set_problem(x0s[b], prob["x_nom"], prob["Q"], prob["R"], prob["Qf"], np.zeros((1, prob["N"] - 1)))
        L, improvement, it = np.inf, np.inf, 0
        while improvement > o.delta and it < 60:
            if it > 0:                         # (the first iteration has no gains: a plain open-loop rollout)
                # which candidates does the reference's line search visit here?
                eps, visited = 1.0, []
                while eps >= 1e-8:
                    visited.append(eps)
                    _, _, Lc, ex = o.rollout(eps)
                    if L - Lc > o.gamma * ex:
                        break
                    eps *= o.beta
                n_iter += 1
                trials_hist.append(len(visited))
                if len(visited) > 1:
                    n_bt += 1
                for pred in (True, False):
                    sw = []
                    for e in visited:
                        s_, defect = newton_rollout(o, e, pred)
                        if s_ < 0:
                            fails[pred] += 1
                            sw.append(None)
                        else:
                            sw.append(s_)
                    stats[pred].append(sw)
            L_new, eps_acc, trials = o.forward(L)
            o.backward()
            improvement = L - L_new
            L = L_new
            it += 1
        print(f"problem {b}: {it} iterations", flush=True)
    th = np.array(trials_hist)
    print(f"\n{n_iter} iterations after the first, {n_bt} of them backtrack ({100.0 * n_bt / n_iter:.0f} %); trials per iteration: mean {th.mean():.2f}, "
          f"histogram {np.bincount(th).tolist()}")
    for pred in (True, False):
        tot = [sum(s for s in sw if s is not None) for sw in stats[pred] if None not in sw]
        per = [s for sw in stats[pred] for s in sw if s is not None]
        nfail_it = sum(1 for sw in stats[pred] if None in sw)
        print(f"start = {'linear predictor' if pred else 'nominal trajectory'}: sweeps per candidate mean {np.mean(per):.2f} median {np.median(per):.0f} "
              f"max {max(per)}; candidates that did not converge in 40 sweeps: {fails[pred]} (in {nfail_it} of {len(stats[pred])} iterations); "
              f"sweeps summed over an iteration's visited candidates: mean {np.mean(tot):.1f} median {np.median(tot):.0f} 90 % {np.percentile(tot, 90):.0f} max {max(tot)}")
        bt = [sum(s for s in sw if s is not None) for sw in stats[pred] if None not in sw and len(sw) > 1]
        if bt:
            print(f"    backtracking iterations only: mean {np.mean(bt):.1f} median {np.median(bt):.0f} 90 % {np.percentile(bt, 90):.0f}")


if __name__ == "__main__":
    main()
