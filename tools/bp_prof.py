import sys, numpy as np
sys.path.insert(0, ".")
from drake_ddp_amd import workloads as W
from drake_ddp_amd.ilqr import BatchedIterativeLQR
from drake_ddp_amd.models import ModelSystem
q = W.synth36_problem(); N = q["N"]
s = BatchedIterativeLQR(ModelSystem(q["model_id"], q["dt"]), N, 1, delta=q["delta"], beta=q["beta"], gamma=q["gamma"], jacobian_mode="ad")
s.SetTargetState(q["x_nom"]); s.SetRunningCost(q["Q"], q["R"]); s.SetTerminalCost(q["Qf"])
s.SetInitialState(W.synth36_batch_x0(64)[:1]); s.SetInitialGuess(W.synth36_u_guess(N))
s.Solve()
c = s.stage_cycles[0]; h = s.history[0, -1, 0]; f = s.history[0, -1, 1]
print("per-step cycles: fetch %.0f  T1 %.0f  H %.0f  factor %.0f  subst %.0f  Vxx+publish %.0f" % (c[0]/39, c[1]/39, c[2]/39, f/39, c[3]/39, h/39))
