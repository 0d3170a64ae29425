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
    "arm27_batch": 0, "arm27_mpc_full": 0, "arm27c_batch": 0, "arm27c_mpc_first": 0,
    "c4_full_leading8": 0,
    # C4 (stiff contact, N = 200): the C oracle itself takes different decisions in 8-9 of the 256 problems when x0 moves by
    # one ulp (tests/test_gpu_keypoints_quad3d_fullsize.py::test_c4_full_size_vs_c_oracle measures that beside this budget)
    "c4_full_history": 8,
}


def assert_flip_budget(name, same, detail=None):
    """`same`: boolean per problem (counts identical to the checker's).  At most FLIP_BUDGET[name] may be False."""
    import numpy as _np
    flipped = int((~_np.asarray(same, bool)).sum())
    assert flipped <= FLIP_BUDGET[name], (name, f"{flipped} problems deviate, budget {FLIP_BUDGET[name]}", detail)


def backward_extended(o):
    """The reference's backward pass (ilqr.py:623-667) on an OracleILQR's current x_bar / u_bar / fx / fu in EXTENDED precision
    (x86 80-bit long double, 64-bit mantissa): the yardstick for a backward pass's round-off - the fp64 NumPy oracle's own
    distance from it says how many digits the PROBLEM leaves (cond(Quu), the dynamic range of Vxx), the device's distance
    from it is then judged against that.  Returns (K (m,n,N-1), kappa (m,N-1), dV (N-1), max cond(Quu))."""
    ld = np.longdouble

    def inv_ld(A):
        k = A.shape[0]
        M_ = np.concatenate([A.astype(ld), np.eye(k, dtype=ld)], axis=1)
        for c in range(k):
            p = c + int(np.argmax(np.abs(M_[c:, c])))
            M_[[c, p]] = M_[[p, c]]
            M_[c] = M_[c] / M_[c, c]
            for r in range(k):
                if r != c:
                    M_[r] = M_[r] - M_[r, c] * M_[c]
        return M_[:, k:]

    Q, R, Qf, x_nom = (np.asarray(a, dtype=ld) for a in (o.Q, o.R, o.Qf, o.x_nom))
    xb, ub, fxa, fua = (np.asarray(a, dtype=ld) for a in (o.x_bar, o.u_bar, o.fx, o.fu))
    n, m, N = o.n, o.m, o.N
    K = np.zeros((m, n, N - 1), dtype=ld); kappa = np.zeros((m, N - 1), dtype=ld); dV = np.zeros(N - 1, dtype=ld)
    Vx = 2 * Qf @ xb[:, -1] - 2 * x_nom @ Qf
    Vxx = 2 * Qf
    cond = 0.0
    for t in range(N - 2, -1, -1):
        fx, fu = fxa[:, :, t], fua[:, :, t]
        lx = 2 * Q @ xb[:, t] - 2 * x_nom @ Q
        lu = 2 * R @ ub[:, t]
        Qx, Qu = lx + fx.T @ Vx, lu + fu.T @ Vx
        Qxx, Quu, Qux = 2 * Q + fx.T @ Vxx @ fx, 2 * R + fu.T @ Vxx @ fu, fu.T @ Vxx @ fx
        cond = max(cond, float(np.linalg.cond(Quu.astype(np.float64))))
        Qi = inv_ld(Quu)
        kappa[:, t] = Qi @ Qu
        K[:, :, t] = Qi @ Qux
        dV[t] = Qu @ Qi @ Qu
        Vx = Qx - Qu @ Qi @ Qux
        Vxx = Qxx - Qux.T @ Qi @ Qux
    return K, kappa, dV, cond


def backward_errors(dev, o):
    """(device's, fp64 oracle's) worst relative distance from the extended-precision backward pass over K, kappa, dV, and
    max cond(Quu).  `dev` = (K, kappa, dV) of the device for the oracle `o`'s inputs; o.backward() must have run."""
    Kx, kx, dx, cond = backward_extended(o)
    e_dev = e_ref = 0.0
    for d_, r_, x_ in zip(dev, (o.K, o.kappa, o.dV), (Kx, kx, dx)):
        sc = float(np.max(np.abs(x_)))
        e_dev = max(e_dev, float(np.max(np.abs(d_ - x_))) / sc)
        e_ref = max(e_ref, float(np.max(np.abs(r_ - x_))) / sc)
    return e_dev, e_ref, cond
