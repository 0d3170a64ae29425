"""Where the time of a Solve() through the class surface goes (C2, B = 1024): python tools/boundary_prof.py"""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import ctypes as C
from drake_ddp_amd import workloads as W, _capi
from test_gpu_parity import make_solver
p = W.pendulum_problem()
x0 = W.pendulum_batch_x0(1024)
for pinned in (False, True):
    s = make_solver(p, B=1024, jac="fd", pinned_results=pinned)
    ug = np.zeros((1, p["N"] - 1))
    T = {}
    def lap(name, t0):
        T[name] = T.get(name, 0.0) + time.perf_counter() - t0
    for r in range(12):
        if r == 2:
            T = {}
        t = time.perf_counter(); s.Reset(); s.SetInitialState(x0); s.SetInitialGuess(ug); lap("setters", t)
        t = time.perf_counter(); s._push_problem(); lap("push (set_cost + set_initial)", t)
        t = time.perf_counter(); st = _capi.Stats(); _capi.check(s._lib.mi_ilqr_solve(s._h, C.byref(st)), "solve"); lap("mi_ilqr_solve", t)
        t = time.perf_counter(); x = s.x_bar; lap("x_bar", t)
        t = time.perf_counter(); u = s.u_bar; lap("u_bar", t)
        t = time.perf_counter(); L = s.cost; lap("cost", t)
    t0 = time.perf_counter()
    for r in range(10):
        s.Reset(); s.SetInitialState(x0); s.SetInitialGuess(ug); s.Solve()
    print("pinned_results", pinned, "Solve() end to end: %.4f ms" % (1e3 * (time.perf_counter() - t0) / 10))
    print("pinned_results", pinned, {k: round(1e3 * v / 10, 4) for k, v in T.items()}, "total ms", round(1e3 * sum(T.values()) / 10, 4))
