"""Shared helpers for the parity tests: golden loading and oracle construction."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    g = {k: z[k] for k in z.files}
    prob = {k[2:]: g[k] for k in g if k.startswith("p_")}
    for k in ("model_id", "N"):
        prob[k] = int(prob[k])
    for k in ("dt", "delta", "beta", "gamma"):
        prob[k] = float(prob[k])
    return g, prob


def golden_keypoint(g):
    if "kp_cfg_method" not in g:
        return None
    nums = g["kp_cfg_nums"]
    return (str(g["kp_cfg_method"]), int(nums[0]), int(nums[1]), float(nums[2]), float(nums[3]))


def make_oracle(prob, keypoint=None, jacobian="ad", fd_step=1e-5):
    from oracle import models_np as M
    from oracle.ilqr_np import OracleILQR, KeypointCfg
    model = M.Model(prob["model_id"], prob["dt"], prob.get("params"))
    kp = KeypointCfg(*keypoint) if keypoint is not None else None
    return OracleILQR(model, prob["N"], delta=prob["delta"], beta=prob["beta"],
                      gamma=prob["gamma"], keypoint=kp, jacobian=jacobian, fd_step=fd_step)


def rel_err(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return float(np.max(np.abs(a - b)) / max(1e-300, np.max(np.abs(b))))


# Problems of a batch whose line-search decisions (iteration / trial counts) differ from the checker's.  An
# ill-conditioned solve can flip one decision at round-off level; the budget of every test is the count OBSERVED on
# its committed seeds (MI355X, round 3) - a regression that makes more problems deviate fails loudly, and a test
# that spends budget says so in its own entry here.
FLIP_BUDGET = {
    "c3_mpc_full": 0, "c5_mpc_full": 0, "quad_mpc_full": 0,
    "randomized_0": 0, "randomized_1": 0, "randomized_2": 0, "randomized_3": 0,
    "short_horizons": 0, "pendulum_mpc_par_vs_seq": 0, "scan_vs_seq_backward": 0,
    "quad_batch_free": 0, "quad_batch_tight": 0, "quad_batch_free_unconverged": 0, "quad_batch_tight_unconverged": 0,
    "quad3d_batch_free": 0, "quad3d_batch_tight": 0, "quad3d_mpc_full": 0,
    "c4_full_leading8": 0,
    # C4 (stiff contact, N = 200): the C oracle itself takes different decisions in 8-9 of the 256 problems when x0 moves by
    # one ulp (tests/test_gpu_round3.py::test_c4_full_size_vs_c_oracle measures that beside this budget)
    "c4_full_history": 8,
}


def assert_flip_budget(name, same, detail=None):
    """`same`: boolean per problem (counts identical to the checker's).  At most FLIP_BUDGET[name] may be False."""
    import numpy as _np
    flipped = int((~_np.asarray(same, bool)).sum())
    assert flipped <= FLIP_BUDGET[name], (name, f"{flipped} problems deviate, budget {FLIP_BUDGET[name]}", detail)
