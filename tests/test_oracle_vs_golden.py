"""Pins the oracle (oracle/ilqr_np.py) against fixtures produced by the
UNMODIFIED reference solver (tests/golden/*.npz, made by oracle/gen_golden.py).
CPU only.  Tolerances: the oracle follows the reference's arithmetic (explicit
inverse, same op order up to NumPy reassociation), so agreement is ~1e-12 rel."""
import numpy as np
import pytest

from common import load_golden, golden_keypoint, make_oracle, rel_err

TOL = 1e-9

SINGLE = ["pendulum_c1", "pendulum_kp_setinterval5", "pendulum_kp_adaptivejerk",
          "pendulum_kp_iterativeerror", "pendulum_c2_00", "pendulum_c2_01", "pendulum_c2_05",
          "acrobot_kp_adaptivejerk", "acrobot_kp_iterativeerror",
          "cartpole_wall_literal_n100", "cartpole_wall_c4_0", "cartpole_plain",
          "quad_solve_0", "quad_infeasible_0", "quad_infeasible_1",
          "quad3d_solve_0", "quad3d_solve_1", "quad3d_infeasible_0", "arm27_solve_0", "arm27_solve_1",
          "arm27_kp_adaptivejerk", "arm27_kp_iterativeerror",
          "arm27c_solve_0", "arm27c_solve_1", "arm27c_kp_adaptivejerk", "arm27c_kp_iterativeerror"]


@pytest.mark.parametrize("name", SINGLE)
def test_single_solve_matches_reference(name):
    g, prob = load_golden(name)
    o = make_oracle(prob, golden_keypoint(g))
    o.set_problem(g["x0"], prob["x_nom"], prob["Q"], prob["R"], prob["Qf"], g["u_guess"])
    x, u, L, hist = o.solve()
    hist = np.array(hist)
    assert len(hist) == len(g["hist"])
    assert np.array_equal(hist[:, 2], g["hist"][:, 2])            # line-search trial counts
    assert np.allclose(hist[:, 1], g["hist"][:, 1], rtol=0, atol=0)  # eps values exact
    assert np.allclose(hist[:, 3], g["hist"][:, 3])               # percentage_derivs
    assert rel_err(hist[:, 0], g["hist"][:, 0]) < TOL
    assert abs(L - g["L"]) <= TOL * abs(g["L"])
    assert np.array_equal(o.keypoints, g["kp_last"])
    for key, val in (("x_bar", x), ("u_bar", u), ("K", o.K), ("kappa", o.kappa),
                     ("dV", o.dV), ("fx", o.fx), ("fu", o.fu)):
        assert rel_err(val, g[key]) < 1e-7, key


@pytest.mark.parametrize("name", ["pendulum_stage", "acrobot_stage", "synth36_stage", "quad_stage", "quad3d_stage", "arm27_stage", "arm27c_stage"])
def test_stage_level(name):
    g, prob = load_golden(name)
    o = make_oracle(prob)
    o.set_problem(g["x0"], prob["x_nom"], prob["Q"], prob["R"], prob["Qf"], g["pre_u_bar"])
    o.x_bar, o.K, o.kappa, o.dV = g["pre_x_bar"].copy(), g["pre_K"].copy(), g["pre_kappa"].copy(), g["pre_dV"].copy()
    x, u, L, _ = o.rollout(1.0)
    assert rel_err(x, g["roll_x"]) < 1e-11 and rel_err(u, g["roll_u"]) < 1e-11
    assert abs(L - g["roll_L"]) < 1e-11 * abs(g["roll_L"])
    o.linearize(x, u)
    assert rel_err(o.fx, g["fx"]) < 1e-12 and rel_err(o.fu, g["fu"]) < 1e-12
    o.x_bar, o.u_bar = x, u
    o.backward()
    assert rel_err(o.K, g["post_K"]) < 1e-10
    assert rel_err(o.kappa, g["post_kappa"]) < 1e-10
    assert rel_err(o.dV, g["post_dV"]) < 1e-10


@pytest.mark.parametrize("name", ["acrobot_mpc_0", "acrobot_mpc_1", "synth36_mpc_0", "quad_mpc_0", "quad3d_mpc_0", "quad3d_mpc_1", "arm27_mpc_0", "arm27c_mpc_0"])
def test_mpc_sequence(name):
    """Receding-horizon re-solves with persistent gains."""
    from drake_ddp_amd.workloads import mpc_shift, synth36_u_guess, planar_quad_u_guess, quad3d_u_guess, arm27_u_guess, arm27c_u_guess
    g, prob = load_golden(name)
    o = make_oracle(prob)
    N, m = prob["N"], g["us"].shape[1]
    u_guess = {4: synth36_u_guess, 5: planar_quad_u_guess, 6: quad3d_u_guess, 7: arm27_u_guess, 8: arm27c_u_guess}.get(prob["model_id"], lambda N_: np.zeros((m, N_ - 1)))(N)
    x0 = g["x0"]
    x_nom = prob["x_nom"].copy()
    replan = int(g["replan"])
    for r in range(len(g["Ls"])):
        if r > 0:
            x0, u_guess = mpc_shift(x, u, replan)
            if "move_target" in g:
                x_nom[int(g["move_target"][0])] += g["move_target"][1]
        o.set_problem(x0, x_nom, prob["Q"], prob["R"], prob["Qf"], u_guess)
        x, u, L, hist = o.solve()
        assert len(hist) == g["iters"][r]
        assert abs(L - g["Ls"][r]) < 1e-9 * abs(g["Ls"][r])
        assert rel_err(x, g["xs"][r]) < 1e-8 and rel_err(u, g["us"][r]) < 1e-7
        assert rel_err(o.K, g["Ks"][r]) < 1e-6


def test_fd_jacobian_close_to_ad():
    """Central FD (the device linearization) vs exact duals; tolerance from SURVEY §8c."""
    from oracle import models_np as M
    rng = np.random.default_rng(5)
    for mid in (0, 1, 2, 3, 4, 5, 6, 7, 8):
        model = M.Model(mid, 0.01)
        x = rng.uniform(-1, 1, model.n)
        u = rng.uniform(-1, 1, model.m)
        if mid == 5:
            x[1] = 0.45 + 0.01 * x[1]                      # trunk height: the feet at the ground, contact active
        if mid == 6:                                       # near the standing state: unit quaternion, feet at the ground
            from oracle import problems as P
            x = P.quad3d_stand() + 0.02 * x
        if mid in (7, 8):                                  # near the start state: the hand at the ball (contact active), the ball on the ground
            from oracle import problems as P
            x = P.arm27_start() + 0.02 * x
            x[0] += 0.04
        fx, fu = model.jac_ad(x, u)
        gx, gu = model.jac_fd(x, u, 1e-5)
        # (quadruped: contact curvature k/sigma^2 = 2.5e8 makes the h^2 truncation term visible; entries reach 1e2)
        # (3-D quadruped: the same curvature through 3-D lever arms and the trunk's small roll inertia: 2e-7 relative)
        # (arm + ball: contact curvature k/sigma^2 = 6e7 on a 0.2 kg ball)
        # (coupled arm: the same contact, this draw leaves the hand 1 mm off the ball - curvature term 6e-7 on entries of order one)
        tol = 2e-9 if mid < 5 else (1e-8 if mid == 5 else (1e-6 if mid == 8 else 2e-7)) * max(1.0, np.max(np.abs(fx)))
        assert np.max(np.abs(fx - gx)) < tol and np.max(np.abs(fu - gu)) < tol


def test_bytes_per_iteration_matches_survey():
    from oracle.ilqr_np import bytes_per_iteration
    assert bytes_per_iteration(2, 1, 200, 1) == 49424
    assert bytes_per_iteration(4, 1, 40, 1) == 22288
    assert bytes_per_iteration(4, 1, 200, 1) == 113168
    assert bytes_per_iteration(36, 12, 40, 1) == 1416704


def test_oracle_and_product_workloads_agree():
    """oracle/problems.py (inputs of the golden generator) and drake_ddp_amd/workloads.py (inputs of
    bench.py / the examples) are written independently from the reference's scripts: bitwise the same
    problems, batches and warm starts."""
    from oracle import problems as P
    from drake_ddp_amd import workloads as W
    for name, args in (("pendulum_problem", ()), ("acrobot_problem", (40,)), ("cartpole_problem", (100,)),
                       ("cartpole_wall_problem", (200,)), ("cartpole_wall_problem", (100,)), ("synth36_problem", (40,)),
                       ("planar_quad_problem", (40,)), ("quad3d_problem", (40,)), ("arm27_problem", (50,)), ("arm27c_problem", (50,))):
        a, b = getattr(P, name)(*args), getattr(W, name)(*args)
        assert a.keys() == b.keys()
        for k in a:
            if isinstance(a[k], np.ndarray):
                assert np.array_equal(a[k], b[k]), (name, k)
            else:
                assert a[k] == b[k], (name, k)
    for name, B in (("pendulum_batch_x0", 1024), ("acrobot_batch_x0", 512), ("cartpole_wall_batch_x0", 256), ("synth36_batch_x0", 64),
                    ("planar_quad_batch_x0", 64), ("quad3d_batch_x0", 64), ("arm27_batch_x0", 64)):
        assert np.array_equal(getattr(P, name)(B), getattr(W, name)(B)), name
    assert np.array_equal(P.synth36_u_guess(40), W.synth36_u_guess(40)) and P.SYNTH_TARGET_VEL == W.SYNTH_TARGET_VEL
    assert np.array_equal(P.quad3d_u_guess(40), W.quad3d_u_guess(40)) and P.QUAD3D_TARGET_VEL == W.QUAD3D_TARGET_VEL
    assert np.array_equal(P.arm27_u_guess(50), W.arm27_u_guess(50)) and np.array_equal(P.arm27_start(), W.arm27_start())
    assert np.array_equal(P.arm27c_u_guess(50), W.arm27c_u_guess(50))
    rng = np.random.default_rng(0)
    x, u = rng.standard_normal((3, 4, 40)), rng.standard_normal((3, 1, 39))
    for got, want in zip(P.mpc_shift(x, u, 2), W.mpc_shift(x, u, 2)):
        assert np.array_equal(got, want)


def test_infeasible_steps_change_the_line_search_like_the_reference():
    """SURVEY F15 (ilqr.py:315-323): quad_infeasible_* were recorded from the unmodified reference with the
    planar quadruped's velocity bound tightened, so that its Drake-shaped update RAISES inside line-search trials.
    The oracle reproduces those histories (test_single_solve_matches_reference); here: the bound really bit
    (with the default bound the same problems take other step sizes) and the planar quadruped's
    articulated-body step conserves energy (free flight, no damping: drift of first order in dt)."""
    from oracle import models_np as M
    for name in ("quad_infeasible_0", "quad_infeasible_1"):
        g, prob = load_golden(name)
        assert prob["params"][8] == 27.5
        free = make_oracle(dict(prob, params=M.DEFAULT_PARAMS[M.PLANAR_QUAD]))
        free.set_problem(g["x0"], prob["x_nom"], prob["Q"], prob["R"], prob["Qf"], g["u_guess"])
        hist = np.array(free.solve()[3])
        assert not np.array_equal(hist[:, 2], g["hist"][:, 2][:len(hist)]) or len(hist) != len(g["hist"])
        assert g["hist"][:, 2].max() >= 2
    # energy of the articulated body in free flight (no contact, no damping, no springs, no actuation)
    p = np.array(M.DEFAULT_PARAMS[M.PLANAR_QUAD], float)
    p[5] = p[6] = p[7] = 0.0
    p[8] = 1e9
    rng = np.random.default_rng(0)
    x = np.zeros(36)
    x[1], x[2] = 5.0, 0.3
    x[3:18] = rng.uniform(-0.8, 0.8, 15)
    x[18:] = rng.uniform(-1, 1, 18)

    def energy(x_):
        q, v = x_[:18], x_[18:]
        bodies = M.quad_bodies()
        mass = [M.QUAD_TRUNK[0]] + [b[3] for b in bodies]
        inert = [M.QUAD_TRUNK[1]] + [b[4] for b in bodies]
        length = [0.0] + [b[2] for b in bodies]
        th, om, pz, vx, vz = [q[2]], [v[2]], [q[1]], [v[0]], [v[1]]
        e = 0.5 * mass[0] * (vx[0] ** 2 + vz[0] ** 2) + 0.5 * inert[0] * om[0] ** 2 + mass[0] * p[0] * pz[0]
        for i in range(1, 16):
            par, at = bodies[i - 1][0], bodies[i - 1][1]
            s_, c_ = np.sin(th[par]), np.cos(th[par])
            dx, dz = (length[par] * s_, -length[par] * c_) if at is None else (c_ * at[0] - s_ * at[1], s_ * at[0] + c_ * at[1])
            th.append(th[par] + q[2 + i]); om.append(om[par] + v[2 + i])
            pz.append(pz[par] + dz); vx.append(vx[par] - om[par] * dz); vz.append(vz[par] + om[par] * dx)
            rx, rz = 0.5 * length[i] * np.sin(th[i]), -0.5 * length[i] * np.cos(th[i])
            cvx, cvz = vx[i] - om[i] * rz, vz[i] + om[i] * rx
            e += 0.5 * mass[i] * (cvx ** 2 + cvz ** 2) + 0.5 * inert[i] * om[i] ** 2 + mass[i] * p[0] * (pz[i] + rz)
        return e

    drift = []
    for dt in (1e-3, 2.5e-4):
        model = M.Model(M.PLANAR_QUAD, dt, p)
        xx, e0 = x.copy(), energy(x)
        for _ in range(int(round(0.1 / dt))):
            xx = model.step(xx, np.zeros(12))
        drift.append(abs(energy(xx) - e0) / abs(e0))
    assert drift[0] < 2e-4 and drift[1] < 0.3 * drift[0]             # first order in dt: the dynamics are consistent


def test_quad3d_model_is_a_rigid_body_with_contact():
    """The build's 3-D quadruped (oracle/models_np.py: quad3d_step; mini_cheetah.py:41-52's state layout).  (i) Free
    flight (no contact, no joint torques or damping): linear momentum falls with g, the body-frame angular momentum
    keeps its norm and the attitude quaternion its length to first order in dt (Euler's equations + the quaternion
    kinematics are consistent).  (ii) The standing state of the workload is an equilibrium of the contact model under
    the standing torques: zero acceleration.  (iii) quad3d_infeasible_0 was recorded with the velocity bound biting:
    the free model takes other step sizes on the same problem."""
    from oracle import models_np as M, problems as P
    p = np.array(M.DEFAULT_PARAMS[M.QUAD3D], float)
    p[5] = 0.0
    p[6] = 1e9
    rng = np.random.default_rng(1)
    x = P.quad3d_stand()
    x[6] = 5.0                                            # far above the ground
    x[19:25] = rng.uniform(-1, 1, 6)
    Ib = p[8:11]
    drift = []
    for dt in (2e-3, 5e-4):
        model = M.Model(M.QUAD3D, dt, p)
        xx = x.copy()
        steps = int(round(0.2 / dt))
        for _ in range(steps):
            xx = model.step(xx, np.zeros(12))
        assert abs(xx[24] - (x[24] - p[0] * 0.2)) < 1e-9 and np.allclose(xx[22:24], x[22:24], atol=1e-12)
        h0, h1 = np.linalg.norm(Ib * x[19:22]), np.linalg.norm(Ib * xx[19:22])
        drift.append((abs(h1 - h0) / h0, abs(np.linalg.norm(xx[0:4]) - 1.0)))
    assert drift[0][0] < 5e-3 and drift[1][0] < 0.4 * drift[0][0]
    assert drift[0][1] < 5e-3 and drift[1][1] < 0.4 * drift[0][1]
    model = M.Model(M.QUAD3D, 4e-3)
    xs = P.quad3d_stand()
    xn = model.step(xs, P.quad3d_u_guess(3)[:, 0])
    # forces and joint torques balance (the standing torques are given to 6 decimals); what is left is the pitch moment
    # of the four feet standing X = 1 cm ahead of their hips (mini_cheetah.py's q0 has the same lean): m g X / Iyy
    rest = np.delete(xn[19:], 1)
    assert np.max(np.abs(rest)) < 2e-7
    X = -(M.Q3_L1 * np.sin(-0.8) + M.Q3_L2 * np.sin(0.8))
    assert abs(xn[20] + 4e-3 * 9.0 * 9.81 * X / 0.26) < 1e-6
    g, prob = load_golden("quad3d_infeasible_0")
    assert prob["params"][6] == 2.5 and g["hist"][:, 2].max() >= 2
    free = make_oracle(dict(prob, params=M.DEFAULT_PARAMS[M.QUAD3D]))
    free.set_problem(g["x0"], prob["x_nom"], prob["Q"], prob["R"], prob["Qf"], g["u_guess"])
    hist = np.array(free.solve()[3])
    assert len(hist) != len(g["hist"]) or not np.array_equal(hist[:, 2], g["hist"][:, 2])
