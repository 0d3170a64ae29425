import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from common import load_golden
from test_gpu_parity import make_solver
mode, jac, single, B = sys.argv[1], sys.argv[2], sys.argv[3] == "1", int(sys.argv[4])
g, prob = load_golden("quad_solve_0")
kw = dict(hist_cap=32) if single else dict(B=B)
s = make_solver(prob, jac=jac, single=single, on_indefinite=mode, **kw)
s.SetInitialState(g["x0"] if single else np.tile(g["x0"], (B, 1))); s.SetInitialGuess(g["u_guess"])
if len(sys.argv) > 5:
    print("stage", sys.argv[5])
    if sys.argv[5] == "forward": s.stage_forward(np.inf)
    if sys.argv[5] == "backward": s.stage_forward(np.inf); s.stage_backward()
    print("ok", s.status)
    sys.exit(0)
x, u, _, L = s.Solve()
print(mode, jac, single, B, "ok", L, s.iterations, s.status)
print("stage_cycles (prof):", [hex(int(v)) for v in s.stage_cycles[0]])
from drake_ddp_amd import _capi
print("trial_cost record:", s._get(_capi.F_TRIAL_COST, (1, 2)))
