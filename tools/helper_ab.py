"""What a second wavefront per problem is worth when it is free: C2's pendulum batch at B = 512 and 256 (one helper / three
helpers share the linearization, every wave still alone on its SIMD) against MI_ILQR_NO_HELPER=1.  At B = 1024 the batch
fills every SIMD with one main wave of 404 registers - a second wave per SIMD does not fit."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
script = """
import sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
from drake_ddp_amd import workloads as W
from test_gpu_parity import make_solver
p = W.pendulum_problem()
for B in (256, 512, 1024):
    s = make_solver(p, B=B, jac="fd")
    s.SetInitialState(W.pendulum_batch_x0(1024)[:B]); s.SetInitialGuess(np.zeros((1, p["N"] - 1)))
    best = 1e9
    for _ in range(30):
        s.Reset(); s.SetInitialGuess(np.zeros((1, p["N"] - 1))); s.Solve(); best = min(best, s.stats.kernel_ms)
    cyc = s.stage_cycles.astype(float); it = s.iterations
    print("B", B, "kernel_ms %%.4f" %% best, "iterations", int(it.sum()), "max", int(it.max()), "M it/s (kernel) %%.1f" %% (it.sum() / best / 1e3),
          "cycles/iter: rollout+ls %%.0f lin %%.0f bp %%.0f" %% ((cyc[:, 0] / it).mean(), (cyc[:, 1] / it).mean(), (cyc[:, 2] / it).mean()))
""" % (ROOT, os.path.join(ROOT, "tests"))
for tag, env in (("helpers (default)", {}), ("MI_ILQR_NO_HELPER=1", {"MI_ILQR_NO_HELPER": "1"})):
    print(tag)
    r = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, env=dict(os.environ, **env))
    print(r.stdout.strip() or r.stderr[-1500:])
