"""Key-point methods at batch scale: wave-per-problem (latency) against lane-per-problem (throughput, KP instantiation of
ilqr_batch.hpp) on 16 384 acrobot problems.  python tools/kp_kernel_ab.py (on the GPU box)."""
import time, numpy as np, sys
sys.path.insert(0, __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))), "tests"))
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from test_gpu_parity import make_solver
from drake_ddp_amd import workloads as W
a = W.acrobot_problem()
B = 16384
x0 = np.tile(W.acrobot_batch_x0(512), (B // 512, 1))
for kp in (None, ("setInterval", 5, 0, 0.0, 0.0), ("adaptiveJerk", 2, 10, 1e-5, 0.0), ("iterativeError", 2, 0, 0.0, 1e-9)):
    for mode in ("latency", "throughput"):
        s = make_solver(a, B=B, keypoint=kp, jac="fd", kernel_mode=mode)
        s.SetInitialState(x0); s.SetInitialGuess(np.zeros((1, a["N"] - 1)))
        s.Solve()
        t = []
        for _ in range(3):
            s.SetInitialGuess(np.zeros((1, a["N"] - 1)))
            t0 = time.perf_counter(); s.Solve(); t.append(time.perf_counter() - t0)
        it = int(s.iterations.sum()); ms = s.last_kernel_ms()
        print(f"{kp[0] if kp else 'setInterval/1':>15} {mode:>10}: kernel {ms:.2f} ms, {it / ms * 1e3 / 1e6:.2f} M it/s, conv {int((s.status==0).sum())}, mean key-points {s.keypoint_count.mean():.1f} of {a['N']-1}")
