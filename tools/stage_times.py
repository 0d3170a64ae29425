"""Per-stage kernel times at the C2 shape (B problems): rollout / forward / linearize / backward / solve."""
import sys, numpy as np
sys.path.insert(0, ".")
from drake_ddp_amd import workloads as W
from drake_ddp_amd.ilqr import BatchedIterativeLQR
from drake_ddp_amd.models import ModelSystem
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
cfg = sys.argv[2] if len(sys.argv) > 2 else "pendulum"
prob = {"pendulum": W.pendulum_problem, "acrobot": W.acrobot_problem, "wall": W.cartpole_wall_problem}[cfg]()
x0 = {"pendulum": W.pendulum_batch_x0, "acrobot": W.acrobot_batch_x0, "wall": W.cartpole_wall_batch_x0}[cfg](B)
N = prob["N"]
s = BatchedIterativeLQR(ModelSystem(prob["model_id"], prob["dt"]), N, B, delta=prob["delta"], beta=prob["beta"], gamma=prob["gamma"], hist_cap=256)
s.SetTargetState(prob["x_nom"]); s.SetRunningCost(prob["Q"], prob["R"]); s.SetTerminalCost(prob["Qf"])
s.SetInitialState(x0); s.SetInitialGuess(np.zeros((1, N - 1)))
s.Solve()
it = s.iterations
print(f"{cfg} B={B} N={N}: solve kernel {s.stats.kernel_ms:.3f} ms, iters mean {it.mean():.2f} max {it.max()}, ls trials {s.ls_trials.sum()}, => {s.stats.kernel_ms*1e3/it.max():.1f} us/iteration (critical problem)")
cyc = s.stage_cycles; crit = int(np.argmax(cyc[:, 3]))
print(f"in-kernel cycles, mean over problems per iteration: ls {cyc[:,0].sum()/it.sum():.0f} lin {cyc[:,1].sum()/it.sum():.0f} bp {cyc[:,2].sum()/it.sum():.0f}; critical problem {crit}: iters {it[crit]} ls_trials {s.ls_trials[crit]} cycles {cyc[crit].tolist()}")
for rep in range(2):
    s.stage_rollout(1.0); t_r = s.last_kernel_ms()
    s.stage_forward(np.inf); t_f = s.last_kernel_ms()
    s.stage_linearize(); t_l = s.last_kernel_ms()
    s.stage_backward(); t_b = s.last_kernel_ms()
print(f"stage kernels (incl. HBM<->LDS staging): rollout {t_r*1e3:.1f} us, forward {t_f*1e3:.1f} us, linearize {t_l*1e3:.1f} us, backward {t_b*1e3:.1f} us")
