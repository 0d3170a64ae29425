import sys, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from common import load_golden, rel_err
from test_gpu_parity import make_solver
g, prob = load_golden("synth36_stage")
s = make_solver(prob, jac="ad")
s.SetInitialState(g["x0"][None]); s.SetInitialGuess(g["roll_u"])
s.set_state(x_bar=g["roll_x"][None], u_bar=g["roll_u"][None], fx=g["fx"][None], fu=g["fu"][None])
s.stage_backward()
K, kap, dV = s.K[0], s.kappa[0], s.dV_coeff[0]
for t in (38, 37, 36, 30, 0):
    print(t, "K", rel_err(K[:, :, t], g["post_K"][:, :, t]), "kappa", rel_err(kap[:, t], g["post_kappa"][:, t]), "dV", abs(dV[t] - g["post_dV"][t]) / abs(g["post_dV"][t]))
