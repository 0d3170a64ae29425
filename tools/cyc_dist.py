"""Per-problem cycles/iteration distribution of the C2 solve kernel (is every wave equally fast?)."""
import sys, numpy as np
sys.path.insert(0, ".")
from drake_ddp_amd import workloads as W
from drake_ddp_amd.ilqr import BatchedIterativeLQR
from drake_ddp_amd.models import ModelSystem
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
prob = W.pendulum_problem(); x0 = W.pendulum_batch_x0(B); N = prob["N"]
s = BatchedIterativeLQR(ModelSystem(prob["model_id"], prob["dt"]), N, B, delta=prob["delta"], beta=prob["beta"], gamma=prob["gamma"])
s.SetTargetState(prob["x_nom"]); s.SetRunningCost(prob["Q"], prob["R"]); s.SetTerminalCost(prob["Qf"])
s.SetInitialState(x0); s.SetInitialGuess(np.zeros((1, N - 1)))
for rep in range(3):
    s.Solve()
    it = s.iterations; cyc = s.stage_cycles
    per = cyc[:, 3] / it
    print(f"rep {rep}: kernel {s.stats.kernel_ms:.3f} ms; total cycles max {cyc[:,3].max()} (iters {it[np.argmax(cyc[:,3])]}); cycles/iter: min {per.min():.0f} p10 {np.percentile(per,10):.0f} median {np.median(per):.0f} p90 {np.percentile(per,90):.0f} max {per.max():.0f}")
    # by position inside the CU-sized group of 4 consecutive blocks and by iteration count
    for k in sorted(set(it.tolist())):
        m = it == k
        print(f"   iters={k:2d}: {m.sum():4d} problems, cycles/iter mean {per[m].mean():.0f} max {per[m].max():.0f}; ls-trials/iter {s.ls_trials[m].mean()/k:.2f}")
    s.reset() if hasattr(s, "reset") else None
    s.SetInitialState(x0); s.SetInitialGuess(np.zeros((1, N - 1)))
