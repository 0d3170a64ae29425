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
          "cartpole_wall_literal_n100", "cartpole_wall_c4_0", "cartpole_plain"]


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


@pytest.mark.parametrize("name", ["pendulum_stage", "acrobot_stage", "synth36_stage"])
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


@pytest.mark.parametrize("name", ["acrobot_mpc_0", "acrobot_mpc_1", "synth36_mpc_0"])
def test_mpc_sequence(name):
    """Receding-horizon re-solves with persistent gains."""
    from drake_ddp_amd.workloads import mpc_shift, synth36_u_guess
    g, prob = load_golden(name)
    o = make_oracle(prob)
    N, m = prob["N"], g["us"].shape[1]
    u_guess = synth36_u_guess(N) if prob["model_id"] == 4 else np.zeros((m, N - 1))
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
    for mid in (0, 1, 2, 3, 4):
        model = M.Model(mid, 0.01)
        x = rng.uniform(-1, 1, model.n)
        u = rng.uniform(-1, 1, model.m)
        fx, fu = model.jac_ad(x, u)
        gx, gu = model.jac_fd(x, u, 1e-5)
        assert np.max(np.abs(fx - gx)) < 2e-9 and np.max(np.abs(fu - gu)) < 2e-9


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
                       ("cartpole_wall_problem", (200,)), ("cartpole_wall_problem", (100,)), ("synth36_problem", (40,))):
        a, b = getattr(P, name)(*args), getattr(W, name)(*args)
        assert a.keys() == b.keys()
        for k in a:
            if isinstance(a[k], np.ndarray):
                assert np.array_equal(a[k], b[k]), (name, k)
            else:
                assert a[k] == b[k], (name, k)
    for name, B in (("pendulum_batch_x0", 1024), ("acrobot_batch_x0", 512), ("cartpole_wall_batch_x0", 256), ("synth36_batch_x0", 64)):
        assert np.array_equal(getattr(P, name)(B), getattr(W, name)(B)), name
    assert np.array_equal(P.synth36_u_guess(40), W.synth36_u_guess(40)) and P.SYNTH_TARGET_VEL == W.SYNTH_TARGET_VEL
    rng = np.random.default_rng(0)
    x, u = rng.standard_normal((3, 4, 40)), rng.standard_normal((3, 1, 39))
    for got, want in zip(P.mpc_shift(x, u, 2), W.mpc_shift(x, u, 2)):
        assert np.array_equal(got, want)
