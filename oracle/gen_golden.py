#!/usr/bin/env python
"""Generate tests/golden/*.npz by running the UNMODIFIED reference solver
(/root/reference/ilqr.py) in THIS container under oracle/pydrake_stub, driven by
the build-owned models (oracle/models_np.py) through oracle/drake_duck.py.

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden.py

Only data (inputs + the reference's outputs) is written; no reference source,
bytecode or derivative of it enters the repo.  /root/reference does not exist on
the GPU box, so nothing at test time imports this script's reference dependency.
"""
import contextlib
import io
import os
import sys

sys.dont_write_bytecode = True      # never write __pycache__ into the mount
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle", "pydrake_stub"))
sys.path.insert(0, "/root/reference")

import numpy as np  # noqa: E402

import ilqr as ref_ilqr  # noqa: E402  (the reference, unmodified)
import utils_derivs_interpolation as ref_utils  # noqa: E402
from oracle import models_np as M  # noqa: E402
from oracle.drake_duck import DuckSystem  # noqa: E402
from oracle import problems as P  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


class Recorder:
    """Wraps a reference solver instance; records per-iteration rows and key-points."""

    def __init__(self, solver):
        self.s = solver
        self.rows = []
        self.kps = []
        fwd, s = solver._forward_pass, solver
        for name in ("get_keypoints_set_interval", "get_keypoints_adaptive_jerk",
                     "get_keypoints_iterative_error"):
            orig = getattr(solver, name)

            def wrapped(*a, _orig=orig):
                kp = _orig(*a)
                self.kps.append(np.array(kp, dtype=np.int64))
                return kp
            setattr(solver, name, wrapped)

        def fwd_wrapped(L_last):
            L, eps, ls = fwd(L_last)
            self.rows.append((L, eps, ls, s.percentage_derivs))
            return L, eps, ls
        solver._forward_pass = fwd_wrapped

    def solve(self):
        self.rows, self.kps = [], []
        with contextlib.redirect_stdout(io.StringIO()):
            x, u, _, L = self.s.Solve()
        s = self.s
        out = dict(x_bar=np.array(x), u_bar=np.array(u), L=float(L),
                   K=s.K.copy(), kappa=s.kappa.copy(), dV=s.dV_coeff.copy(),
                   fx=s.fx.copy(), fu=s.fu.copy(),
                   hist=np.array(self.rows, dtype=float),
                   kp_last=self.kps[-1].copy(),
                   kp_all=np.concatenate(self.kps), kp_len=np.array([len(k) for k in self.kps]))
        return out


def make_ref(prob, keypoint=None):
    model = M.Model(prob["model_id"], prob["dt"], prob.get("params"))
    kp = None
    if keypoint is not None:
        kp = ref_utils.derivs_interpolation(*keypoint)
    s = ref_ilqr.IterativeLinearQuadraticRegulator(
        DuckSystem(model), prob["N"], delta=prob["delta"], beta=prob["beta"],
        gamma=prob["gamma"], derivs_keypoint_method=kp)
    s.SetTargetState(prob["x_nom"])
    s.SetRunningCost(prob["Q"], prob["R"])
    s.SetTerminalCost(prob["Qf"])
    return s


def save(name, prob, **arrays):
    meta = {k: np.asarray(v) for k, v in prob.items() if k not in ("name",) and v is not None}
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **{"p_" + k: v for k, v in meta.items()},
                        **arrays)
    info = ""
    if "hist" in arrays:
        h = arrays["hist"]
        info = f"iters={len(h)} L={arrays['L']:.6g} max_ls={int(h[:,2].max())}"
    if "iters" in arrays:
        info = f"iters={arrays['iters'].tolist()} Ls={np.round(arrays['Ls'],4).tolist()}"
    print("wrote", name, info)


def flat(prefix, d):
    return {f"{prefix}{k}": v for k, v in d.items()}


def single_solve(name, prob, x0, u_guess, keypoint=None):
    s = make_ref(prob, keypoint)
    s.SetInitialState(np.array(x0, float))
    s.SetInitialGuess(np.array(u_guess, float))
    out = Recorder(s).solve()
    extra = {}
    if keypoint is not None:
        extra["kp_cfg_method"] = np.array(keypoint[0])
        extra["kp_cfg_nums"] = np.array(keypoint[1:], dtype=float)
    save(name, prob, x0=np.array(x0, float), u_guess=np.array(u_guess, float), **out, **extra)
    return out


def stage_level(name, prob, x0, u_guess, n_iters=3):
    """Drive _forward_pass/_backward_pass by hand (ilqr.py:695-697) and snapshot
    the solver state before/after each stage of the LAST driven iteration."""
    s = make_ref(prob)
    s.SetInitialState(np.array(x0, float))
    s.SetInitialGuess(np.array(u_guess, float))
    L = np.inf
    with contextlib.redirect_stdout(io.StringIO()):
        for _ in range(n_iters - 1):
            L, _, _ = s._forward_pass(L)
            s._backward_pass()
        pre = dict(x_bar=s.x_bar.copy(), u_bar=s.u_bar.copy(), K=s.K.copy(),
                   kappa=s.kappa.copy(), dV=s.dV_coeff.copy())
        eps, x, u, Lr, ls = s._linesearch(np.inf)          # eps = 1 rollout body
        s._get_derivatives(x, u)
        s.u_bar, s.x_bar = u, x
        s._backward_pass()
    save(name, prob, x0=np.array(x0, float), **flat("pre_", pre), L_last=float(L),
         roll_x=x, roll_u=u, roll_L=float(Lr), roll_eps=float(eps),
         fx=s.fx.copy(), fu=s.fu.copy(),
         post_K=s.K.copy(), post_kappa=s.kappa.copy(), post_dV=s.dV_coeff.copy())


def mpc(name, prob, x0, u_guess, resolves, replan, move_target=None):
    """MPC loop exactly as acrobot.py:142-155 / mini_cheetah.py:186-201: shift the
    control tape by `replan`, repeat the last column, restart from x[:,replan];
    gains persist inside the solver between solves."""
    s = make_ref(prob)
    rec = Recorder(s)
    x_nom = np.array(prob["x_nom"], float)
    xs, us, Ls, iters, Ks = [], [], [], [], []
    x0 = np.array(x0, float)
    u_guess = np.array(u_guess, float)
    for r in range(resolves + 1):
        if r > 0:
            last = u[:, -1]
            u_guess = np.block([u[:, replan:], np.repeat(last[np.newaxis].T, replan, axis=1)])
            x0 = x[:, replan]
            if move_target is not None:
                idx, inc = move_target
                x_nom[idx] += inc
                s.SetTargetState(x_nom)
        s.SetInitialState(x0)
        s.SetInitialGuess(u_guess)
        out = rec.solve()
        x, u = out["x_bar"], out["u_bar"]
        xs.append(x.copy()); us.append(u.copy()); Ls.append(out["L"])
        iters.append(len(out["hist"])); Ks.append(out["K"].copy())
    extra = {}
    if move_target is not None:
        extra["move_target"] = np.array(move_target, dtype=float)
    save(name, prob, x0=np.array(xs[0][:, 0]), u_guess0=np.zeros(0), xs=np.array(xs), us=np.array(us),
         Ls=np.array(Ls), iters=np.array(iters), Ks=np.array(Ks),
         replan=np.array(replan), **extra)


def main():
    os.makedirs(OUT, exist_ok=True)

    # C1: pendulum.py literal (pendulum.py:18-34, 84-98)
    c1 = P.pendulum_problem()
    single_solve("pendulum_c1", c1, [0.0, 0.0], np.zeros((1, c1["N"] - 1)))
    stage_level("pendulum_stage", c1, [0.0, 0.0], np.zeros((1, c1["N"] - 1)))

    # key-point modes on C1 (parameters as the survey probe)
    single_solve("pendulum_kp_setinterval5", c1, [0.0, 0.0], np.zeros((1, 199)),
                 keypoint=("setInterval", 5, 0, 0.0, 0.0))
    single_solve("pendulum_kp_adaptivejerk", c1, [0.0, 0.0], np.zeros((1, 199)),
                 keypoint=("adaptiveJerk", 5, 20, 1e-4, 0.0))
    single_solve("pendulum_kp_iterativeerror", c1, [0.0, 0.0], np.zeros((1, 199)),
                 keypoint=("iterativeError", 5, 0, 0.0, 5e-5))

    # C2: first problems of the B=1024 batch (rng seed 0)
    x0s = P.pendulum_batch_x0(1024)
    for i in range(12):
        single_solve(f"pendulum_c2_{i:02d}", c1, x0s[i], np.zeros((1, 199)))

    # C3: acrobot-shaped MPC, N=40, first 2 problems of the seed-1 batch, 6 resolves
    c3 = P.acrobot_problem()
    ax0 = P.acrobot_batch_x0(512)
    for i in range(2):
        mpc(f"acrobot_mpc_{i}", c3, ax0[i], np.zeros((1, c3["N"] - 1)), resolves=6, replan=2)
    stage_level("acrobot_stage", c3, ax0[0], np.zeros((1, c3["N"] - 1)), n_iters=2)
    single_solve("acrobot_kp_adaptivejerk", c3, ax0[0], np.zeros((1, c3["N"] - 1)),
                 keypoint=("adaptiveJerk", 2, 10, 1e-5, 0.0))
    single_solve("acrobot_kp_iterativeerror", c3, ax0[0], np.zeros((1, c3["N"] - 1)),
                 keypoint=("iterativeError", 2, 0, 0.0, 1e-9))

    # C4: cart-pole with wall, N=200 (config) and the script-literal N=100
    c4 = P.cartpole_wall_problem(N=200)
    wx0 = P.cartpole_wall_batch_x0(256)
    single_solve("cartpole_wall_literal_n100", P.cartpole_wall_problem(N=100),
                 [0.0, np.pi + 0.5, 0.0, 0.0], np.zeros((1, 99)))
    for i in range(2):
        single_solve(f"cartpole_wall_c4_{i}", c4, wx0[i], np.zeros((1, 199)))
    single_solve("cartpole_plain", P.cartpole_problem(), [0.0, np.pi - 0.6, 0.0, 0.0], np.zeros((1, 99)))

    # C5: cheetah-shaped synthetic, N=40, MPC with moving target
    c5 = P.synth36_problem()
    sx0 = P.synth36_batch_x0(64)
    ug = P.synth36_u_guess(c5["N"])
    mpc("synth36_mpc_0", c5, sx0[0], ug, resolves=2, replan=4,
        move_target=(0, P.SYNTH_TARGET_VEL * c5["dt"] * 4))
    stage_level("synth36_stage", c5, sx0[0], ug, n_iters=2)

    # (f)4: the planar quadruped (articulated-body dynamics + ground contact), C5's shape on a physical model.
    # quad_infeasible_*: |v| bound tightened to 27.5 so that line-search trials are declared INFEASIBLE by the
    # model (RuntimeError out of the Drake-shaped update, caught at ilqr.py:315-323: L = inf) and the accepted
    # step sizes differ from the unconstrained run's.
    cq = P.planar_quad_problem()
    qx0 = P.planar_quad_batch_x0(8)
    qug = P.planar_quad_u_guess(cq["N"])
    stage_level("quad_stage", cq, qx0[0], qug, n_iters=2)
    single_solve("quad_solve_0", cq, qx0[0], qug)
    mpc("quad_mpc_0", cq, qx0[1], qug, resolves=2, replan=4, move_target=(0, P.QUAD_TARGET_VEL * cq["dt"] * 4))
    tight = np.array(M.DEFAULT_PARAMS[M.PLANAR_QUAD], float)
    tight[8] = 27.5
    cqt = dict(cq, params=tight)
    for k, i in enumerate((6, 7)):
        single_solve(f"quad_infeasible_{k}", cqt, qx0[i], qug)

    # (f)4: the 3-D quadruped (quaternion floating base, n = 37, feet contact): mini_cheetah.py's own state layout,
    # cost weights, moving target (x_nom[4] += target_vel * dt * replan, :151-156) and MPC loop (:186-201).
    c3d = P.quad3d_problem()
    x3 = P.quad3d_batch_x0(8)
    u3 = P.quad3d_u_guess(c3d["N"])
    stage_level("quad3d_stage", c3d, x3[0], u3, n_iters=2)
    single_solve("quad3d_solve_0", c3d, x3[0], u3)
    mpc("quad3d_mpc_0", c3d, x3[2], u3, resolves=3, replan=4, move_target=(4, P.QUAD3D_TARGET_VEL * c3d["dt"] * 4))
    # the reference's literal target velocity (mini_cheetah.py:25: 1.0 m/s): line searches that backtrack, and - with the
    # velocity bound tightened to 2.5 - trials the model declares infeasible (ilqr.py:315-323)
    c3f = P.quad3d_problem(target_vel=1.0)
    single_solve("quad3d_solve_1", c3f, x3[1], u3)
    mpc("quad3d_mpc_1", c3f, x3[2], u3, resolves=3, replan=4, move_target=(4, 1.0 * c3f["dt"] * 4))
    tight3 = np.array(M.DEFAULT_PARAMS[M.QUAD3D], float)
    tight3[6] = 2.5
    single_solve("quad3d_infeasible_0", dict(c3f, params=tight3), x3[3], u3)

    # (f)4, the mid-size shape: a 7-joint arm pushing a free ball - the state kinova_gen3.py:52-70 stacks (n = 27, m = 7),
    # its cost (:73-87), horizon (T = 0.5, dt = 1e-2), delta = 1e-3, beta = 0.5 (:258-259) and gravity-compensation initial
    # guess (:268-275); one receding-horizon re-solve sequence in the style of the MPC scripts
    ca = P.arm27_problem()
    xa = P.arm27_batch_x0(8)
    ua = P.arm27_u_guess(ca["N"])
    stage_level("arm27_stage", ca, P.arm27_start(), ua, n_iters=3)
    single_solve("arm27_solve_0", ca, P.arm27_start(), ua)
    single_solve("arm27_solve_1", ca, xa[1], ua)
    mpc("arm27_mpc_0", ca, xa[2], ua, resolves=2, replan=5)
    # kinova_gen3.py:34-40's derivative interpolation on the same problem (adaptiveJerk with its literal minN = 5, maxN = 40,
    # jerk threshold 1e-4 - dof = int(27 / 2) = 13 "velocity rows" x[13:26], ilqr.py:444,477-484 - and iterativeError, minN = 5)
    single_solve("arm27_kp_adaptivejerk", ca, xa[3], ua, keypoint=("adaptiveJerk", 5, 40, 1e-4, 0.0))
    single_solve("arm27_kp_iterativeerror", ca, xa[3], ua, keypoint=("iterativeError", 5, 0, 0.0, 1e-2))

    # (f)4 widened (round 5): the same arm, ball, cost and horizon with COUPLED rigid-body joint dynamics (model 8: joint-space mass
    # matrix, centripetal / Coriolis and gravity terms of three point masses + rotor inertias) - the manipulator kinova_gen3.py:105-213
    # builds from its URDF is a coupled chain, ARM27's joints are not
    cc = P.arm27c_problem()
    uc = P.arm27c_u_guess(cc["N"])
    stage_level("arm27c_stage", cc, P.arm27_start(), uc, n_iters=3)
    single_solve("arm27c_solve_0", cc, P.arm27_start(), uc)
    single_solve("arm27c_solve_1", cc, xa[1], uc)
    mpc("arm27c_mpc_0", cc, xa[2], uc, resolves=2, replan=5)
    single_solve("arm27c_kp_adaptivejerk", cc, xa[3], uc, keypoint=("adaptiveJerk", 5, 40, 1e-4, 0.0))
    single_solve("arm27c_kp_iterativeerror", cc, xa[3], uc, keypoint=("iterativeError", 5, 0, 0.0, 1e-2))

if __name__ == "__main__":
    main()
