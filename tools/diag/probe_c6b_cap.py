import sys, os, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MI_ILQR_LIB"] = os.path.join(ROOT, "drake_ddp_amd/lib/dbg/libmi_cap.so")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from drake_ddp_amd import workloads as W, _capi
from test_gpu_parity import make_solver
p = W.arm27c_problem()
B = int(os.environ.get("PROBE_B", "64"))
s = make_solver(p, B=B, jac="fd"); s.SetInitialState(W.arm27_batch_x0(B)); s.SetInitialGuess(W.arm27c_u_guess(p["N"]))
t0 = time.time(); s.Solve(); print("cold solve %.2f s iters %d status %s" % (time.time() - t0, s.iterations.sum(), np.unique(s.status)), flush=True)
t0 = time.time()
try:
    s.MPCRun(1, 5)
except Exception as e:
    print("MPCRun raised", repr(e)[:200])
print("MPCRun(1) %.2f s iters %s status %s" % (time.time() - t0, s.iterations[:16], s.status[:16]), flush=True)
w = np.empty((B, 8), dtype=np.uint64)
_capi.check(s._lib.mi_ilqr_get_int(s._h, _capi.I64_CLUSTER_WORDS, _capi.ptr(w), w.nbytes), "get")
for b in range(min(B, 6)):
    print(b, "cmd round %d flags %x parts %d | done %d | alive %d | exit rounds %d same %d | progress round %d val %x | early open %d hit %d | ls %d" % (
        w[b,0] >> 32, (w[b,0] >> 16) & 0xffff, w[b,0] & 0xffff, w[b,1], w[b,2] & 0xffff, w[b,3] >> 32, (w[b,3] >> 8) & 0xffffff, w[b,4] >> 32, w[b,4] & 0xffffffff, w[b,5] >> 32, w[b,5] & 0xffffffff, s.ls_trials[b]))
print("mpc_log iters", s.mpc_log[:6, :, -1].ravel())
