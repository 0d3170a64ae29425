"""ctypes binding of libmi_ilqr.so (include/mi_ilqr.h).  Plumbing only.

The library is the product: if it is missing or fails to load this module raises
— there is no Python/NumPy fallback for any compute entry.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# (MI_ILQR_LIB: another build of the same library, e.g. the AddressSanitizer one of drake_ddp_amd/build.py)
LIB_PATH = os.environ.get("MI_ILQR_LIB") or os.path.join(_HERE, "lib", "libmi_ilqr.so")

MAX_PARAMS = 16
ABI_VERSION = 9

# enums (include/mi_ilqr.h)
OK, E_BAD_SHAPE, E_BAD_METHOD, E_LINESEARCH, E_HIP, E_NO_DEVICE, E_BAD_ARG, E_UNSUPPORTED, E_RCCL = 0, -1, -2, -3, -4, -5, -6, -7, -8
COMM_ID_BYTES, COMM_MAX_COUNT = 128, 64
KP_SET_INTERVAL, KP_ADAPTIVE_JERK, KP_ITERATIVE_ERROR = 0, 1, 2
JAC_FD_CENTRAL, JAC_AUTODIFF = 0, 1
KERNEL_AUTO, KERNEL_LATENCY, KERNEL_THROUGHPUT = 0, 1, 2
STATUS_CONVERGED, STATUS_MAX_ITERS, STATUS_LINESEARCH_FAILED, STATUS_INTERNAL = 0, 1, 2, 3
STATUS_NOT_PD = 5
STATUS_FLAG_INDEFINITE = 16     # OR-ed onto the outcome: on_indefinite="continue" inverted a Quu that is not positive definite
F_X_BAR, F_U_BAR, F_K, F_KAPPA, F_DV, F_FX, F_FU, F_COST, F_X0, F_HIST, F_X_TRIAL, F_U_TRIAL, F_TRIAL_COST, F_ITER_CYCLES = range(14)
I_ITERS, I_STATUS, I_LS_TRIALS, I_KP_COUNT, I_KP_LIST = 100, 101, 102, 103, 104
I64_STAGE_CYCLES = 200
I64_CLUSTER_WORDS = 201
CLUSTER_WORDS = 40          # MI_ILQR_CLUSTER_WORDS

EXPORTS = [
    "mi_ilqr_abi_version", "mi_ilqr_struct_sizes", "mi_ilqr_strerror", "mi_ilqr_model_info", "mi_ilqr_register_model", "mi_ilqr_create", "mi_ilqr_destroy",
    "mi_ilqr_set_cost", "mi_ilqr_set_initial", "mi_ilqr_set_initial_shared", "mi_ilqr_host_alloc", "mi_ilqr_host_free", "mi_ilqr_set_result_sink", "mi_ilqr_solve_into", "mi_ilqr_reset", "mi_ilqr_rearm_initial_guess",
    "mi_ilqr_solve", "mi_ilqr_solve_async", "mi_ilqr_collect_stats", "mi_ilqr_collect_stats_n",
    "mi_ilqr_rollout", "mi_ilqr_forward", "mi_ilqr_linearize", "mi_ilqr_backward", "mi_ilqr_mpc_shift",
    "mi_ilqr_mpc_run", "mi_ilqr_get_mpc_log",
    "mi_ilqr_get", "mi_ilqr_get_int", "mi_ilqr_get_async", "mi_ilqr_set", "mi_ilqr_device_ptr", "mi_ilqr_get_stream",
    "mi_ilqr_synchronize", "mi_ilqr_last_kernel_ms", "mi_ilqr_set_timing", "mi_ilqr_get_cycles", "mi_ilqr_bytes_per_iteration", "mi_ilqr_lds_bytes",
    "mi_ilqr_comm_unique_id", "mi_ilqr_comm_create", "mi_ilqr_comm_destroy", "mi_ilqr_comm_count", "mi_ilqr_allreduce_min",
    "mi_ilqr_allreduce_min_start", "mi_ilqr_allreduce_min_wait",
]


class Desc(C.Structure):
    _fields_ = [
        ("n", C.c_int32), ("m", C.c_int32), ("N", C.c_int32), ("B", C.c_int32),
        ("model_id", C.c_int32), ("n_params", C.c_int32),
        ("model_params", C.c_double * MAX_PARAMS),
        ("dt", C.c_double), ("delta", C.c_double), ("beta", C.c_double), ("gamma", C.c_double),
        ("keypoint_method", C.c_int32), ("minN", C.c_int32), ("maxN", C.c_int32),
        ("jerk_threshold", C.c_double), ("iterative_error_threshold", C.c_double),
        ("jacobian_mode", C.c_int32), ("fd_step", C.c_double),
        ("max_iters", C.c_int32), ("hist_cap", C.c_int32), ("device_id", C.c_int32), ("kernel_mode", C.c_int32),
        ("on_indefinite", C.c_int32),
    ]


class Stats(C.Structure):
    _fields_ = [
        ("total_iters", C.c_int64), ("total_ls_trials", C.c_int64),
        ("n_converged", C.c_int32), ("n_max_iters", C.c_int32), ("n_ls_failed", C.c_int32),
        ("max_iters_seen", C.c_int32),
        ("best_cost", C.c_double), ("best_index", C.c_int32), ("kernel_ms", C.c_float),
        ("algorithmic_bytes", C.c_double), ("n_internal", C.c_int32), ("n_not_pd", C.c_int32),
    ]


_lib = None


def load():
    """Load libmi_ilqr.so; raises ImportError loudly when it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is not built. Run `python -m drake_ddp_amd.build` (hipcc, gfx950). "
            "There is no CPU fallback for the iLQR hot path.")
    lib = C.CDLL(LIB_PATH)
    H = C.c_void_p
    dp = C.POINTER(C.c_double)
    lib.mi_ilqr_abi_version.restype = C.c_int
    lib.mi_ilqr_strerror.restype = C.c_char_p
    lib.mi_ilqr_strerror.argtypes = [C.c_int]
    lib.mi_ilqr_model_info.argtypes = [C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32), dp]
    lib.mi_ilqr_register_model.argtypes = [C.c_void_p, C.POINTER(C.c_int32)]
    lib.mi_ilqr_create.argtypes = [C.POINTER(Desc), C.POINTER(H)]
    lib.mi_ilqr_destroy.argtypes = [H]
    lib.mi_ilqr_destroy.restype = None
    lib.mi_ilqr_set_cost.argtypes = [H, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.mi_ilqr_set_initial.argtypes = [H, C.c_void_p, C.c_void_p]
    for name in ("mi_ilqr_reset", "mi_ilqr_rearm_initial_guess", "mi_ilqr_solve_async", "mi_ilqr_linearize",
                 "mi_ilqr_backward", "mi_ilqr_synchronize"):
        getattr(lib, name).argtypes = [H]
    lib.mi_ilqr_solve.argtypes = [H, C.POINTER(Stats)]
    lib.mi_ilqr_collect_stats.argtypes = [H, C.POINTER(Stats)]
    lib.mi_ilqr_collect_stats_n.argtypes = [H, C.c_int32, C.POINTER(Stats)]
    lib.mi_ilqr_rollout.argtypes = [H, C.c_void_p]
    lib.mi_ilqr_forward.argtypes = [H, C.c_void_p]
    lib.mi_ilqr_set_initial_shared.argtypes = [H, C.c_void_p, C.c_void_p]
    lib.mi_ilqr_host_alloc.argtypes = [C.c_size_t, C.POINTER(C.c_void_p)]
    lib.mi_ilqr_host_free.argtypes = [C.c_void_p]
    lib.mi_ilqr_set_result_sink.argtypes = [H, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.mi_ilqr_solve_into.argtypes = [H, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.POINTER(Stats), C.POINTER(C.c_int32)]
    lib.mi_ilqr_mpc_shift.argtypes = [H, C.c_int32]
    lib.mi_ilqr_mpc_run.argtypes = [H, C.c_int32, C.c_int32, C.c_void_p, C.POINTER(Stats)]
    lib.mi_ilqr_get_mpc_log.argtypes = [H, C.c_void_p, C.c_size_t]
    lib.mi_ilqr_get.argtypes = [H, C.c_int, C.c_void_p, C.c_size_t]
    lib.mi_ilqr_get_int.argtypes = [H, C.c_int, C.c_void_p, C.c_size_t]
    lib.mi_ilqr_get_async.argtypes = [H, C.c_int, C.c_void_p, C.c_size_t]
    lib.mi_ilqr_set.argtypes = [H, C.c_int, C.c_void_p, C.c_size_t]
    lib.mi_ilqr_device_ptr.argtypes = [H, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
    lib.mi_ilqr_get_stream.argtypes = [H, C.POINTER(C.c_void_p)]
    lib.mi_ilqr_last_kernel_ms.argtypes = [H, C.POINTER(C.c_float)]
    lib.mi_ilqr_set_timing.argtypes = [H, C.c_int32]
    lib.mi_ilqr_get_cycles.argtypes = [H, C.c_void_p, C.c_size_t]
    lib.mi_ilqr_bytes_per_iteration.restype = C.c_double
    lib.mi_ilqr_bytes_per_iteration.argtypes = [C.c_int32] * 4
    lib.mi_ilqr_lds_bytes.restype = C.c_size_t
    lib.mi_ilqr_lds_bytes.argtypes = [C.POINTER(Desc)]
    lib.mi_ilqr_comm_unique_id.argtypes = [C.c_void_p]
    lib.mi_ilqr_comm_create.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(H)]
    lib.mi_ilqr_comm_destroy.argtypes = [H]
    lib.mi_ilqr_comm_destroy.restype = None
    lib.mi_ilqr_comm_count.argtypes = [H, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    lib.mi_ilqr_allreduce_min.argtypes = [H, C.c_void_p, C.c_int32]
    lib.mi_ilqr_allreduce_min_start.argtypes = [H, C.c_void_p, C.c_int32]
    lib.mi_ilqr_allreduce_min_wait.argtypes = [H, C.c_void_p, C.c_int32]
    if lib.mi_ilqr_abi_version() != ABI_VERSION:
        raise ImportError("libmi_ilqr.so ABI version mismatch")
    # this module restates mi_ilqr_desc / mi_ilqr_stats with ctypes: their sizes must be the library's
    d_, s_ = C.c_int32(), C.c_int32()
    lib.mi_ilqr_struct_sizes.restype = None
    lib.mi_ilqr_struct_sizes(C.byref(d_), C.byref(s_), None)
    if (d_.value, s_.value) != (C.sizeof(Desc), C.sizeof(Stats)):
        raise ImportError(f"libmi_ilqr.so struct layout mismatch: desc {d_.value} / stats {s_.value} bytes in the library, "
                          f"{C.sizeof(Desc)} / {C.sizeof(Stats)} here")
    _lib = lib
    return lib


class MiIlqrError(RuntimeError):
    def __init__(self, code, where):
        self.code = code
        msg = load().mi_ilqr_strerror(code).decode()
        super().__init__(f"{where}: {msg} (code {code})")


def check(code, where):
    if code != OK:
        raise MiIlqrError(code, where)


def as_f64(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None and a.shape != tuple(shape):
        raise AssertionError(f"expected shape {tuple(shape)}, got {a.shape}")
    return a


def ptr(a):
    """Address of an array's first element (an int: ctypes takes it for a c_void_p argument), None for None."""
    return a.__array_interface__["data"][0] if a is not None else None
