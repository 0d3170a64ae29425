"""The backward pass as its OWN kernel (MODE_BACKWARD: mi_ilqr stage API) against the same pass inside the fused solve kernel
(in-kernel stopwatch): what the fused kernel's register allocation costs the pass.  B = 1 and B = 64; cycles at the shader clock."""
import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from drake_ddp_amd import workloads as W
from test_gpu_parity import make_solver

CASES = [("synth36", W.synth36_problem(), W.synth36_batch_x0, W.synth36_u_guess),
         ("planar_quad", W.planar_quad_problem(), W.planar_quad_batch_x0, W.planar_quad_u_guess),
         ("quad3d", W.quad3d_problem(), W.quad3d_batch_x0, W.quad3d_u_guess),
         ("arm27", W.arm27_problem(), W.arm27_batch_x0, W.arm27_u_guess),
         ("arm27c", W.arm27c_problem(), W.arm27_batch_x0, W.arm27c_u_guess)]
for name, p, x0f, ugf in CASES:
    for B in (1, 64):
        s = make_solver(p, B=B, jac="fd")
        s.SetInitialState(x0f(64)[:B]); s.SetInitialGuess(ugf(p["N"]))
        s.Solve()
        cyc = s.stage_cycles.astype(float); it = s.iterations
        fused = (cyc[:, 2] / it).mean()
        best = 1e9
        for rep in range(5):
            s.stage_backward(); best = min(best, s.last_kernel_ms())
        clk = fused / 1.0
        print(f"{name:12s} B {B:3d}: fused-kernel backward pass {fused:9.0f} cycles ({fused / (p['N'] - 1):6.0f} per step); standalone kernel {best * 1e3:8.1f} us "
              f"= {best * 1e-3 * 2.4e9:9.0f} cycles at 2.4 GHz ({best * 1e-3 * 2.4e9 / (p['N'] - 1):6.0f} per step, launch included)", flush=True)
