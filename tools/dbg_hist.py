import sys, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from common import load_golden
from test_gpu_parity import make_solver
name = sys.argv[1]; jac = sys.argv[2] if len(sys.argv) > 2 else "ad"
g, prob = load_golden(name)
s = make_solver(prob, jac=jac, single=True, hist_cap=128)
s.SetInitialState(g["x0"]); s.SetInitialGuess(g["u_guess"])
x, u, _, L = s.Solve()
it = int(s.iterations[0]); h = s.history[0][:it]; gh = g["hist"]
for i in range(max(it, len(gh))):
    a = h[i] if i < it else [np.nan]*4; b = gh[i] if i < len(gh) else [np.nan]*4
    print(f"{i:3d} gpu L={a[0]:.12g} eps={a[1]:.4g} ls={a[2]:.0f} | ref L={b[0]:.12g} eps={b[1]:.4g} ls={b[2]:.0f} | rel={abs(a[0]-b[0])/abs(b[0]):.2e}")
