"""Where the host time of a single-problem Solve() through the drop-in class goes (C1): python tools/diag/c1_profile.py"""
import cProfile, pstats, io, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from drake_ddp_amd import workloads as W
from drake_ddp_amd.ilqr import IterativeLinearQuadraticRegulator
from drake_ddp_amd.models import ModelSystem
p = W.pendulum_problem()
ilqr = IterativeLinearQuadraticRegulator(ModelSystem(p["model_id"], p["dt"]), p["N"], delta=p["delta"], beta=p["beta"], gamma=p["gamma"], verbose=False)
ilqr.SetTargetState(p["x_nom"]); ilqr.SetRunningCost(p["Q"], p["R"]); ilqr.SetTerminalCost(p["Qf"])
u0 = np.zeros((1, p["N"] - 1))
def once():
    ilqr.Reset()
    t0 = time.perf_counter()
    ilqr.SetInitialState(np.zeros(2)); ilqr.SetInitialGuess(u0)
    x, u, _, L = ilqr.Solve()
    return time.perf_counter() - t0
for _ in range(5): once()
ts = [once() for _ in range(200)]
print("median %.4f ms  min %.4f ms  kernel %.4f ms" % (1e3 * np.median(ts), 1e3 * min(ts), ilqr.stats.kernel_ms))
pr = cProfile.Profile(); pr.enable()
for _ in range(300): once()
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22); print(s.getvalue()[:4500])
