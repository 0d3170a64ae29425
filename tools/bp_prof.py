"""Phase profile of the workgroup-per-problem backward pass (csrc/ilqr_large.hpp: large_backward, fused chain).

Builds its own -DMI_PROF_BACKWARD library (lib/libmi_ilqr_bpprof.so: per-wave stopwatches between the phases, thread 0 =
a matrix-core wave, thread 192 = the solver wave) and prints cycles per step.

    python tools/bp_prof.py [--light] [synth36|quad|quad3d]
"""
import os, sys
sys.path.insert(0, ".")
from drake_ddp_amd import build as B
light = "--light" in sys.argv
if light: sys.argv.remove("--light")
lib = os.path.join(B.LIBDIR, "libmi_ilqr_bpprof%s.so" % ("_light" if light else ""))
B.build(verbose=False, extra=["-DMI_PROF_BACKWARD"] + (["-DMI_PROF_BACKWARD_LIGHT"] if light else []), lib=lib)
os.environ["MI_ILQR_LIB"] = lib
import numpy as np
from drake_ddp_amd import workloads as W
from drake_ddp_amd.ilqr import BatchedIterativeLQR
from drake_ddp_amd.models import ModelSystem

which = sys.argv[1] if len(sys.argv) > 1 else "synth36"
if which == "quad3d":
    q = W.quad3d_problem(); x0 = W.quad3d_batch_x0(1); ug = W.quad3d_u_guess(q["N"])
elif which == "quad":
    q = W.planar_quad_problem(); x0 = W.planar_quad_batch_x0(1); ug = W.planar_quad_u_guess(q["N"])
else:
    q = W.synth36_problem(); x0 = W.synth36_batch_x0(64)[:1]; ug = W.synth36_u_guess(q["N"])
N = q["N"]
s = BatchedIterativeLQR(ModelSystem(q["model_id"], q["dt"]), N, 1, delta=q["delta"], beta=q["beta"], gamma=q["gamma"], jacobian_mode="ad")
s.SetTargetState(q["x_nom"]); s.SetRunningCost(q["Q"], q["R"]); s.SetTerminalCost(q["Qf"])
s.SetInitialState(x0); s.SetInitialGuess(ug)
s.Solve()
H = s.history[0]; cap = H.shape[0]
st = N - 1
w0 = H[cap - 4:cap].reshape(-1) / st; w3 = H[cap - 8:cap - 4].reshape(-1) / st
print(f"{which}: n = {q['x_nom'].size}, {st} steps; cycles per step")
if light:
    print("light mode (stopwatches at the barriers only): first half-step busy | wait || second half-step busy | wait")
    print("  matrix-core wave 0: %.0f | %.0f || %.0f | %.0f   (sum %.0f)" % (w0[2], w0[3], w0[12], w0[11], w0[[2, 3, 12, 11]].sum()))
    print("  solver wave:        %.0f | %.0f || %.0f | %.0f   (sum %.0f)" % (w3[6], w3[8], w3[7], w3[11], w3[[6, 8, 7, 11]].sum()))
    sys.exit(0)
print("matrix-core wave 0: T1 = Vxx F %.0f | H = F^T T1 + Qux store %.0f | wait %.0f || K, Vxx' + stores %.0f | share of the next Quu %.0f | wait %.0f   (sum %.0f)"
      % (w0[1], w0[2], w0[3], w0[4], w0[12], w0[11], w0[[0, 1, 2, 3, 4, 11, 12]].sum()))
print("solver wave:        gather Quu %.0f | Gauss-Jordan + store %.0f | wait %.0f || kappa, dV, Vx' %.0f | publish F, prefetch %.0f | next first-order column %.0f | wait %.0f   (sum %.0f)"
      % (w3[5], w3[6], w3[8], w3[9], w3[10], w3[7], w3[11], w3[[0, 5, 6, 7, 8, 9, 10, 11]].sum()))
print("(the stopwatches themselves cost ~25 %: the release build runs the 36-state chain at ~7.1 k cycles per step)")
print("prologue, cycles per backward pass (wave 0): init stores + terminal Vx %.0f  cost gradients %.0f  first F fetch+publish %.0f"
      % (st * w0[13], st * w0[14], st * w0[15]))
