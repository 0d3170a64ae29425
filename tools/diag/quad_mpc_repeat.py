import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from drake_ddp_amd import workloads as W
from test_gpu_parity import make_solver
B = 4
q, x0, ug = W.planar_quad_problem(), W.planar_quad_batch_x0(B), W.planar_quad_u_guess(40)
for rep in range(4):
    s = make_solver(q, B=B, jac="fd")
    s.SetInitialState(x0); s.SetInitialGuess(ug)
    s.Solve()
    c0 = s.cost.copy()
    st = s.MPCRun(1, 4)
    print(rep, "cold", c0.round(6).tolist(), "-> status", s.status.tolist(), "iters", s.iterations.tolist(), "trials", s.ls_trials.tolist(), "cost", s.cost.round(5).tolist(),
          "hist row0", s.history[0, 0].round(5).tolist())
# host-loop form of the same re-solve (MODE_SOLVE after a shift): what the first re-solve should be
s = make_solver(q, B=B, jac="fd")
s.SetInitialState(x0); s.SetInitialGuess(ug); s.Solve()
s.MPCShift(4); st = s.solve_resident()
print("host loop: status", s.status.tolist(), "iters", s.iterations.tolist(), "cost", s.cost.round(5).tolist(), "hist row0", s.history[0, 0].round(5).tolist())
