import sys, numpy as np
sys.path.insert(0, ".")
from drake_ddp_amd import workloads as W
from drake_ddp_amd.ilqr import BatchedIterativeLQR
from drake_ddp_amd.models import ModelSystem
prob = W.pendulum_problem(); N = prob["N"]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
for label, x0 in (("random x0", W.pendulum_batch_x0(B)), ("identical x0 (no divergence)", np.repeat(W.pendulum_batch_x0(4)[1:2], B, axis=0)),
                  ("sorted by iteration count", None)):
    if x0 is None:
        x0 = W.pendulum_batch_x0(B)
        x0 = x0[np.argsort(prev_iters, kind="stable")]
    s = BatchedIterativeLQR(ModelSystem(prob["model_id"], prob["dt"]), N, B, delta=prob["delta"], beta=prob["beta"], gamma=prob["gamma"], kernel_mode="throughput", hist_cap=2)
    s.SetTargetState(prob["x_nom"]); s.SetRunningCost(prob["Q"], prob["R"]); s.SetTerminalCost(prob["Qf"])
    s.SetInitialState(x0); s.SetInitialGuess(np.zeros((1, N - 1))); s._push_problem()
    s.rearm(); s.solve_resident(); s.rearm(); st = s.solve_resident()
    if label == "random x0": prev_iters = s.iterations.copy()
    print(f"{label:32s} B={B} kernel {st.kernel_ms:8.3f} ms  iters total {st.total_iters} (mean {st.total_iters/B:.2f}, max {st.max_iters_seen})  {st.total_iters/st.kernel_ms*1e3:.3e} it/s  -> {st.kernel_ms*1e3/ (st.max_iters_seen):.1f} us per max-iteration", flush=True)
    del s
