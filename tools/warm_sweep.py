"""C2 (B = 1024) in groups of 20 pipelined cold-start solves, from an idle GPU: how the step time settles as the
shader clock ramps up, with HIP events on every launch and on one launch in four.   python tools/warm_sweep.py"""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from drake_ddp_amd import workloads as W
from test_gpu_parity import make_solver
p = W.pendulum_problem()
for every in (1, 4, 0):
    s = make_solver(p, B=1024, jac="fd")
    s.SetInitialState(W.pendulum_batch_x0(1024)); s.SetInitialGuess(np.zeros((1, p["N"] - 1))); s._push_problem()
    s.set_timing(every)
    time.sleep(2.0)                      # let the device fall back to idle clocks
    for rep in range(10):
        t0 = time.perf_counter()
        for _ in range(20):
            s.rearm(cold=True); s.solve_resident_async()
        st = s.collect(20)
        dt = time.perf_counter() - t0
        k = [x.kernel_ms for x in st if x.kernel_ms > 0]
        print("events on one launch in %d, group %2d: %.4f ms/step  %.2f M it/s  kernel %.4f ms" %
              (every, rep, 1e3 * dt / 20, sum(x.total_iters for x in st) / dt / 1e6, sum(k) / max(1, len(k))))
    del s
