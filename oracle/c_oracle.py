"""ctypes loader for oracle/lib/libilqr_oracle.so (test infrastructure / CPU baseline)."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "lib", "libilqr_oracle.so")


class Cfg(C.Structure):
    _fields_ = [("n", C.c_int), ("m", C.c_int), ("N", C.c_int), ("model_id", C.c_int),
                ("params", C.c_double * 16), ("dt", C.c_double), ("delta", C.c_double), ("beta", C.c_double),
                ("gamma", C.c_double), ("minN", C.c_int), ("fd_h", C.c_double), ("max_iters", C.c_int),
                ("kp_method", C.c_int), ("maxN", C.c_int), ("jerk_thr", C.c_double), ("err_thr", C.c_double)]


KP_METHODS = {"setInterval": 0, "adaptiveJerk": 1, "iterativeError": 2}


NATIVE_FLAGS = ["-O3", "-march=native", "-ffp-contract=fast", "-fno-fast-math", "-fPIC", "-shared", "-fopenmp", "-std=c11"]
_native = None


def load(native=False):
    """The checker build (oracle/Makefile: -O2, no contraction - bit-comparable with the NumPy restatement), or with
    native=True a build tuned for THIS host's cores (NATIVE_FLAGS, compiled now into a temporary directory: a
    -march=native object must not travel between machines) - the one bench.py times as the CPU baseline."""
    global _native
    if native:
        if _native is None:
            import tempfile
            out = os.path.join(tempfile.mkdtemp(prefix="ilqr_oracle_native_"), "libilqr_oracle_native.so")
            subprocess.check_call([os.environ.get("CC", "gcc")] + NATIVE_FLAGS + [os.path.join(HERE, "ilqr_oracle.c"), "-o", out, "-lm"])
            _native = C.CDLL(out)
            _native.oracle_solve_batch.restype = C.c_int
            _native.oracle_mpc_batch.restype = C.c_int
            _native.oracle_solve_batch_ex.restype = C.c_int
        return _native
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(os.path.join(HERE, "ilqr_oracle.c")):
        subprocess.check_call(["make", "-C", HERE, "-s"])
    lib = C.CDLL(LIB)
    lib.oracle_solve_batch.restype = C.c_int
    lib.oracle_mpc_batch.restype = C.c_int
    lib.oracle_solve_batch_ex.restype = C.c_int
    return lib


def solve_batch(model, prob, x0, u_guess=None, minN=1, fd_h=1e-5, nthreads=0, want_arrays=True, native=False,
                keypoint=None, hist_cap=0, max_iters=100000):
    """Cold-start batched solve on the host.  model: oracle.models_np.Model.  keypoint: (method, minN, maxN,
    jerk_threshold, iterative_error_threshold) like utils_derivs_interpolation.derivs_interpolation; hist_cap > 0
    also returns hist (B, hist_cap, 4) = cost | eps | trials | key-point count per iteration, and the key-points of
    the last linearization (kp_count (B,), kp_list (B, N-1))."""
    lib = load(native)
    n, m, N = model.n, model.m, prob["N"]
    x0 = np.ascontiguousarray(x0, dtype=np.float64).reshape(-1, n)
    B = x0.shape[0]
    cfg = Cfg(n=n, m=m, N=N, model_id=model.model_id, dt=model.dt, delta=prob["delta"], beta=prob["beta"],
              gamma=prob["gamma"], minN=minN, fd_h=fd_h, max_iters=int(max_iters))
    if keypoint is not None:
        cfg.kp_method, cfg.minN, cfg.maxN = KP_METHODS[keypoint[0]], int(keypoint[1]), int(keypoint[2])
        cfg.jerk_thr, cfg.err_thr = float(keypoint[3]), float(keypoint[4])
    for i, v in enumerate(model.params):
        cfg.params[i] = float(v)
    f = lambda a: np.ascontiguousarray(a, dtype=np.float64)
    Q, R, Qf, xn = f(prob["Q"]), f(prob["R"]), f(prob["Qf"]), f(prob["x_nom"])
    ug = None if u_guess is None else f(np.broadcast_to(u_guess, (B, m, N - 1)))
    out = dict(cost=np.empty(B), iters=np.empty(B, np.int32), ls=np.empty(B, np.int32), status=np.empty(B, np.int32))
    if want_arrays:
        out.update(x_bar=np.empty((B, n, N)), u_bar=np.empty((B, m, N - 1)), K=np.empty((B, m, n, N - 1)),
                   kappa=np.empty((B, m, N - 1)))
    if hist_cap > 0:
        out.update(hist=np.zeros((B, hist_cap, 4)), kp_count=np.zeros(B, np.int32), kp_list=np.zeros((B, N - 1), np.int32))
    p = lambda a: a.ctypes.data_as(C.c_void_p) if a is not None else None
    used = lib.oracle_solve_batch_ex(C.byref(cfg), B, p(Q), p(R), p(Qf), p(xn), p(x0), p(ug),
                                     p(out.get("x_bar")), p(out.get("u_bar")), p(out.get("K")), p(out.get("kappa")),
                                     p(out["cost"]), p(out["iters"]), p(out["ls"]), p(out["status"]), int(nthreads),
                                     p(out.get("hist")), int(hist_cap), p(out.get("kp_count")), p(out.get("kp_list")))
    out["threads"] = used
    return out


def mpc_batch(model, prob, x0, u_guess, resolves, replan, target_step=None, minN=1, fd_h=1e-5, nthreads=0):
    """Cold solve + `resolves` receding-horizon re-solves per problem with the solver state persisting
    (acrobot.py:131-162, mini_cheetah.py:186-213).  Returns log (B,resolves,n+2), first (B,2) and the
    final arrays."""
    lib = load()
    n, m, N = model.n, model.m, prob["N"]
    x0 = np.ascontiguousarray(x0, dtype=np.float64).reshape(-1, n)
    B = x0.shape[0]
    cfg = Cfg(n=n, m=m, N=N, model_id=model.model_id, dt=model.dt, delta=prob["delta"], beta=prob["beta"],
              gamma=prob["gamma"], minN=minN, fd_h=fd_h, max_iters=100000)
    for i, v in enumerate(model.params):
        cfg.params[i] = float(v)
    f = lambda a: np.ascontiguousarray(a, dtype=np.float64)
    Q, R, Qf, xn = f(prob["Q"]), f(prob["R"]), f(prob["Qf"]), f(prob["x_nom"])
    ug = None if u_guess is None else f(np.broadcast_to(u_guess, (B, m, N - 1)))
    ts = None if target_step is None else f(target_step)
    out = dict(log=np.empty((B, resolves, n + 2)), first=np.empty((B, 2)), x_bar=np.empty((B, n, N)),
               u_bar=np.empty((B, m, N - 1)), K=np.empty((B, m, n, N - 1)), kappa=np.empty((B, m, N - 1)),
               ls=np.empty(B, np.int32), status=np.empty(B, np.int32))
    p = lambda a: a.ctypes.data_as(C.c_void_p) if a is not None else None
    used = lib.oracle_mpc_batch(C.byref(cfg), B, p(Q), p(R), p(Qf), p(xn), p(x0), p(ug), int(resolves), int(replan), p(ts),
                                p(out["log"]), p(out["first"]), p(out["x_bar"]), p(out["u_bar"]), p(out["K"]), p(out["kappa"]),
                                p(out["ls"]), p(out["status"]), int(nthreads))
    assert used > 0
    out["threads"] = used
    return out
