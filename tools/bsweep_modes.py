"""C2 batch sweep of the two small-state kernel families (wave-per-problem vs lane-per-problem): kernel time, it/s, roofline fraction."""
import sys, time, numpy as np
sys.path.insert(0, ".")
from drake_ddp_amd import workloads as W
from drake_ddp_amd.ilqr import BatchedIterativeLQR
from drake_ddp_amd.models import ModelSystem
prob = W.pendulum_problem(); N = prob["N"]
for mode, Bs in (("latency", (1024, 2048, 4096, 16384, 65536, 262144)), ("throughput", (4096, 16384, 65536, 262144))):
    for B in Bs:
        x0 = W.pendulum_batch_x0(B)
        s = BatchedIterativeLQR(ModelSystem(prob["model_id"], prob["dt"]), N, B, delta=prob["delta"], beta=prob["beta"], gamma=prob["gamma"], kernel_mode=mode, hist_cap=2)
        s.SetTargetState(prob["x_nom"]); s.SetRunningCost(prob["Q"], prob["R"]); s.SetTerminalCost(prob["Qf"])
        s.SetInitialState(x0); s.SetInitialGuess(np.zeros((1, N - 1))); s._push_problem()
        for _ in range(4 if B <= 16384 else 1):              # (bring the clock up: small launches are short)
            s.rearm(); s.solve_resident()
        s.rearm(); st = s.solve_resident()
        gbps = st.algorithmic_bytes / (st.kernel_ms * 1e-3) / 1e9
        print(f"{mode:10s} B={B:7d} kernel {st.kernel_ms:9.3f} ms  {st.total_iters/st.kernel_ms*1e3:.3e} it/s  algorithmic {gbps:8.1f} GB/s ({gbps/8000:.3f} of 8 TB/s)  max iters {st.max_iters_seen}", flush=True)
        del s
