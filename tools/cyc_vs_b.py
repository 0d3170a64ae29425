"""Per-iteration stage cycles of the C2 solve kernel against the batch size."""
import sys, numpy as np
sys.path.insert(0, ".")
from drake_ddp_amd import workloads as W
from drake_ddp_amd.ilqr import BatchedIterativeLQR
from drake_ddp_amd.models import ModelSystem
prob = W.pendulum_problem(); N = prob["N"]
for B in (1, 16, 64, 256, 512, 1024, 1280, 2048):
    x0 = W.pendulum_batch_x0(B)
    s = BatchedIterativeLQR(ModelSystem(prob["model_id"], prob["dt"]), N, B, delta=prob["delta"], beta=prob["beta"], gamma=prob["gamma"])
    s.SetTargetState(prob["x_nom"]); s.SetRunningCost(prob["Q"], prob["R"]); s.SetTerminalCost(prob["Qf"])
    s.SetInitialState(x0); s.SetInitialGuess(np.zeros((1, N - 1)))
    s.Solve(); s.rearm(); s.solve_resident()
    it = s.iterations; cyc = s.stage_cycles; ls = s.ls_trials
    print(f"B={B:5d} kernel {s.stats.kernel_ms*1e3:8.1f} us  per-iteration cycles: rollout/trial {cyc[:,0].sum()/ls.sum():8.0f} ({cyc[:,0].sum()/ls.sum()/(N-1):.0f}/step)  lin {cyc[:,1].sum()/it.sum():7.0f}  bp {cyc[:,2].sum()/it.sum():8.0f} ({cyc[:,2].sum()/it.sum()/(N-1):.0f}/step)  it/s {it.sum()/s.stats.kernel_ms*1e3:.3e}")
