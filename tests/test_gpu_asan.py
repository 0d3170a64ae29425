"""SURVEY.md section 5 (race / memory-error detection): the host side of the C ABI under AddressSanitizer.
drake_ddp_amd/build.py:build_asan links csrc/mi_ilqr.hip compiled with -fsanitize=address (host code; gfx950 device
code cannot be instrumented) against the regular kernel objects; a child process preloads the sanitizer runtime and
drives every kind of entry point - create / set / solve (blocking, pipelined, result sink) / stage calls / MPC on the
device and through the host loop / layout-converting get and set / plugin registration / RCCL communicator / destroy.
Any heap overflow, use-after-free or double free in the host code aborts the child."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import ctypes, sys, numpy as np
# (the HIP runtime is brought up BEFORE the instrumented library registers its code objects: under the sanitizer's
#  interposed loader the other order ends in a null call inside libamdhip64 - nothing of ours is on that stack)
_hip = ctypes.CDLL("libamdhip64.so"); _n = ctypes.c_int(); assert _hip.hipGetDeviceCount(ctypes.byref(_n)) == 0
sys.path.insert(0, %(root)r); sys.path.insert(0, %(root)r + "/tests"); sys.path.insert(0, %(root)r + "/examples/plugins")
from drake_ddp_amd import workloads as W, _capi
assert "asan" in _capi.LIB_PATH
from test_gpu_parity import make_solver
p = W.pendulum_problem()
for pinned in (False, True):
    s = make_solver(p, B=70, jac="fd", pinned_results=pinned)
    s.SetInitialState(W.pendulum_batch_x0(1024)[:70]); s.SetInitialGuess(np.zeros((1, p["N"] - 1)))
    x, u, _, L = s.Solve()
    for _ in range(40):                               # more pipelined solves than the statistics ring holds
        s.rearm(); s.solve_resident_async()
    st = s.collect(32)
    assert st[-1].n_converged == 70
    s.stage_rollout(0.5); s.stage_linearize(); s.stage_backward()
    s.MPCRun(3, 2)
    _ = s.K, s.fx, s.keypoint_list, s.history, s.stage_cycles, s.mpc_log
    del s
a = W.acrobot_problem(N=750)                           # lane-per-problem layout: host-loop MPC, relayouts through scratch
s = make_solver(a, B=3, jac="ad")
s.SetInitialState(np.zeros((3, 4))); s.SetInitialGuess(np.zeros((1, 749)))
s.Solve(); s.MPCRun(2, 5, target_step=np.array([0.0, 0.0, 0.0, 0.0]))
_ = s.x_bar, s.K; s.set_state(x_bar=s.x_bar)
del s
q = W.quad3d_problem()                                 # workgroup-per-problem layout, cluster words, time-major conversions
s = make_solver(q, B=5, jac="fd")
s.SetInitialState(W.quad3d_batch_x0(5)); s.SetInitialGuess(W.quad3d_u_guess(q["N"]))
s.Solve(); s.MPCRun(2, 4); _ = s.fx, s.fu; s.set_state(K=s.K)
del s
import models as PM
mk = PM.build_all()
from drake_ddp_amd.ilqr import BatchedIterativeLQR
v = BatchedIterativeLQR(mk["vdp"](0.02), 50, 9)
v.SetTargetState(np.zeros(2)); v.SetInitialState(np.ones((9, 2))); v.SetInitialGuess(np.zeros((1, 49))); v.Solve()
del v
from drake_ddp_amd.dist import NativeComm
c = NativeComm(0, 1, 0); assert c.allreduce_min([3.0, 1.0])[1] == 1.0; del c
print("ASAN_RUN_OK")
"""


@pytest.mark.gpu
def test_host_side_under_address_sanitizer():
    from drake_ddp_amd import build
    if not os.path.exists(build.ASAN_LIB) or not os.path.exists(build.ASAN_RT):
        pytest.skip("libmi_ilqr_asan.so / the sanitizer runtime is not here (python -m drake_ddp_amd.build --asan)")
    env = dict(os.environ, LD_PRELOAD=build.ASAN_RT, MI_ILQR_LIB=build.ASAN_LIB,
               ASAN_OPTIONS="detect_leaks=0:verify_asan_link_order=0:abort_on_error=1:protect_shadow_gap=0:allocator_may_return_null=1")
    r = subprocess.run([sys.executable, "-c", SCRIPT % {"root": ROOT}], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "ASAN_RUN_OK" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])
    assert "ERROR: AddressSanitizer" not in r.stderr
