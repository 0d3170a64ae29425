"""Lane-per-problem ("throughput") solves of the acrobot at N = 40 for the rocprofv3 passes of tools/pmc_throughput.py and for
bench.py's `throughput` entry:  python tools/run_throughput.py B [kp]   (kp: none | adaptiveJerk | iterativeError | setInterval5)
Three cold batched solves from resident inputs; prints one JSON line about the last."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from drake_ddp_amd import workloads as W
from drake_ddp_amd.ilqr import BatchedIterativeLQR
from drake_ddp_amd.models import ModelSystem
from drake_ddp_amd.utils_derivs_interpolation import derivs_interpolation

KP = {"none": None, "setInterval5": ("setInterval", 5, 0, 0.0, 0.0), "adaptiveJerk": ("adaptiveJerk", 2, 10, 1e-5, 0.0),
      "iterativeError": ("iterativeError", 2, 0, 0.0, 1e-9)}


def make(B, kp="none", device=0):
    a = W.acrobot_problem()
    k = KP[kp]
    s = BatchedIterativeLQR(ModelSystem(a["model_id"], a["dt"]), a["N"], B, delta=a["delta"], beta=a["beta"], gamma=a["gamma"],
                            derivs_keypoint_method=derivs_interpolation(*k) if k else None, jacobian_mode="fd", kernel_mode="throughput",
                            hist_cap=2, device=device)
    s.SetTargetState(a["x_nom"]); s.SetRunningCost(a["Q"], a["R"]); s.SetTerminalCost(a["Qf"])
    x0 = np.tile(W.acrobot_batch_x0(512), (max(1, B // 512), 1))[:B]
    s.SetInitialState(x0); s.SetInitialGuess(np.zeros((1, a["N"] - 1))); s._push_problem()
    return a, s


if __name__ == "__main__":
    B = int(sys.argv[1]); kp = sys.argv[2] if len(sys.argv) > 2 else "none"
    a, s = make(B, kp)
    for _ in range(3):
        s.rearm(cold=True); st = s.solve_resident()
    print(json.dumps({"B": B, "kp": kp, "N": a["N"], "kernel_ms": st.kernel_ms, "iterations": int(st.total_iters), "ls_trials": int(st.total_ls_trials),
                      "converged": int(st.n_converged), "algorithmic_bytes": st.algorithmic_bytes,
                      "mean_keypoints": float(s.keypoint_count.mean())}))
