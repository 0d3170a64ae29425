"""Phase profile of the workgroup-per-problem backward pass (needs a -DMI_PROF_BACKWARD build): cycles per phase A/B/C/D."""
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
H = s.history[0]; cap = H.shape[0]
w0 = H[cap - 4:cap].reshape(-1) / 39; w3 = H[cap - 8:cap - 4].reshape(-1) / 39
print("matrix-core wave 0, cycles/step: top %.0f  A %.0f  B %.0f  C %.0f  D-work %.0f  D-wait %.0f   (sum %.0f)" % (w0[0], w0[1], w0[2], w0[3], w0[12], w0[4], w0[[0,1,2,3,4,12]].sum()))
print("solver wave 3,      cycles/step: top %.0f  A: fetch %.0f  F^T Vx %.0f  back-subst(t+1) %.0f  wait %.0f | B: Quu tile %.0f  LDL %.0f  wait %.0f | C fwd-subst %.0f | D: Vx %.0f  publish %.0f  wait %.0f" % (w3[0], w3[5], w3[6], w3[9], w3[1], w3[7], w3[8], w3[2], w3[3], w3[10], w3[11], w3[4]))
print("prologue, cycles per backward pass (wave 0): init stores + terminal Vx %.0f  cost gradients %.0f  first F fetch+publish %.0f  rest %.0f" % (39*w0[13], 39*w0[14], 39*w0[15], 39*w0[0]))
print("  cost gradients split (wave 0): staging %.0f  dot products %.0f  barrier wait %.0f  zeroing+sync %.0f" % (39*w0[5], 39*w0[6], 39*w0[7], 39*w0[14]))
