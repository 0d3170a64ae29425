"""Stage cycle counters of the workgroup-per-problem kernel at the C5 shape."""
import sys, numpy as np
sys.path.insert(0, ".")
from drake_ddp_amd import workloads as W
from drake_ddp_amd.ilqr import BatchedIterativeLQR
from drake_ddp_amd.models import ModelSystem
q = W.synth36_problem(); N = q["N"]
for B in (1, 8, 64):
  for jac in ("fd", "ad"):
    s = BatchedIterativeLQR(ModelSystem(q["model_id"], q["dt"]), N, B, delta=q["delta"], beta=q["beta"], gamma=q["gamma"], jacobian_mode=jac)
    s.SetTargetState(q["x_nom"]); s.SetRunningCost(q["Q"], q["R"]); s.SetTerminalCost(q["Qf"])
    s.SetInitialState(W.synth36_batch_x0(64)[:B]); s.SetInitialGuess(W.synth36_u_guess(N))
    s.Solve(); s.rearm(); s.solve_resident()
    it = s.iterations; cyc = s.stage_cycles; ls = s.ls_trials
    print(f"B={B} {jac}: kernel {s.stats.kernel_ms*1e3:.0f} us iters {it.tolist()[:8]} ls {ls.tolist()[:8]}  cycles/trial rollout {cyc[:,0].sum()/ls.sum():.0f} ({cyc[:,0].sum()/ls.sum()/(N-1):.0f}/step)  lin/iter {cyc[:,1].sum()/it.sum():.0f}  bp/iter {cyc[:,2].sum()/it.sum():.0f} ({cyc[:,2].sum()/it.sum()/(N-1):.0f}/step)")
