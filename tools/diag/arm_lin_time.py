import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from drake_ddp_amd import workloads as W
from test_gpu_parity import make_solver
p = W.arm27_problem()
for B in (256, 1024):
    s = make_solver(p, B=B, jac="fd")
    s.SetInitialState(W.arm27_batch_x0(B)); s.SetInitialGuess(W.arm27_u_guess(p["N"]))
    s.stage_forward(np.inf)
    ms = []
    for _ in range(5):
        s.stage_linearize(); ms.append(s.last_kernel_ms())
    print(os.environ.get("MI_ILQR_DENSE_JAC", "0"), "B", B, "linearize kernel ms", ["%.3f" % m for m in ms])
