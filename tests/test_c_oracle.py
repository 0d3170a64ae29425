"""Pins the C restatement (oracle/ilqr_oracle.c — the reported CPU baseline) against the
NumPy oracle (same FD Jacobians; ~1e-9) and against the reference's goldens (exact Jacobians
there, so FD tolerance).  CPU only."""
import numpy as np
import pytest

from common import load_golden, make_oracle, rel_err


@pytest.mark.parametrize("name", ["pendulum_c1", "pendulum_c2_01", "pendulum_kp_setinterval5", "acrobot_kp_adaptivejerk"])
def test_c_oracle_matches_numpy_oracle(name):
    from oracle import c_oracle, models_np as M
    g, prob = load_golden(name)
    minN = 5 if "setinterval5" in name else 1
    kp = ("setInterval", minN, 0, 0.0, 0.0)
    o = make_oracle(prob, keypoint=kp, jacobian="fd", fd_step=1e-5)
    o.set_problem(g["x0"], prob["x_nom"], prob["Q"], prob["R"], prob["Qf"], g["u_guess"])
    x, u, L, hist = o.solve()
    r = c_oracle.solve_batch(M.Model(prob["model_id"], prob["dt"]), prob, g["x0"][None], g["u_guess"], minN=minN)
    assert r["iters"][0] == len(hist) and r["ls"][0] == sum(h[2] for h in hist) and r["status"][0] == 0
    assert abs(r["cost"][0] - L) < 1e-10 * abs(L)
    # the acrobot optimum is flat (cost agrees to 1e-10 while x moves by 1e-7): FD round-off level
    tx = 1e-6 if prob["model_id"] == 1 else 1e-9
    assert rel_err(r["x_bar"][0], x) < tx and rel_err(r["u_bar"][0], u) < 10 * tx
    assert rel_err(r["K"][0], o.K) < 100 * tx
    # kappa -> 0 at the optimum: what is left is finite-difference noise, compare absolutely
    assert np.max(np.abs(r["kappa"][0] - o.kappa)) < (1e-4 if prob["model_id"] == 1 else 1e-7)


def test_c_oracle_vs_reference_golden_and_threads():
    from oracle import c_oracle, models_np as M
    names = [f"pendulum_c2_{i:02d}" for i in range(8)]
    gs = [load_golden(n) for n in names]
    prob = gs[0][1]
    x0 = np.stack([g["x0"] for g, _ in gs])
    r1 = c_oracle.solve_batch(M.Model(prob["model_id"], prob["dt"]), prob, x0, np.zeros((1, prob["N"] - 1)), nthreads=1)
    r4 = c_oracle.solve_batch(M.Model(prob["model_id"], prob["dt"]), prob, x0, np.zeros((1, prob["N"] - 1)), nthreads=4)
    assert r1["threads"] == 1 and r4["threads"] == 4
    assert np.array_equal(r1["cost"], r4["cost"]) and np.array_equal(r1["x_bar"], r4["x_bar"])   # threading is bitwise neutral
    for b, (g, _) in enumerate(gs):
        assert r1["iters"][b] == len(g["hist"])
        assert abs(r1["cost"][b] - g["L"]) < 1e-8 * abs(g["L"])
        assert np.max(np.abs(r1["x_bar"][b] - g["x_bar"])) < 1e-6
        assert rel_err(r1["K"][b], g["K"]) < 1e-5
