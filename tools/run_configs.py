"""Run the five BASELINE.json configs on one GPU and print one JSON line each
(iterations/s, ms/solve, algorithmic GB/s).  Not the bench contract (bench.py is)."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drake_ddp_amd import workloads as W
from drake_ddp_amd.ilqr import BatchedIterativeLQR
from drake_ddp_amd.models import ModelSystem


def make(prob, B, **kw):
    s = BatchedIterativeLQR(ModelSystem(prob["model_id"], prob["dt"]), prob["N"], B, delta=prob["delta"],
                            beta=prob["beta"], gamma=prob["gamma"], hist_cap=8, **kw)
    s.SetTargetState(prob["x_nom"]); s.SetRunningCost(prob["Q"], prob["R"]); s.SetTerminalCost(prob["Qf"])
    return s


def single(name, prob, x0, u_guess, reps=5, **kw):
    B = len(x0)
    s = make(prob, B, **kw)
    s.SetInitialState(x0); s.SetInitialGuess(u_guess); s._push_problem()
    for _ in range(2):
        s.rearm(); s.solve_resident()
    t0 = time.perf_counter(); it = 0; kms = 0; ab = 0
    for _ in range(reps):
        s.rearm(); st = s.solve_resident(); it += st.total_iters; kms += st.kernel_ms; ab += st.algorithmic_bytes
    dt = time.perf_counter() - t0
    print(json.dumps({"config": name, "B": B, "iters_per_solve": it / reps, "max_iters": st.max_iters_seen,
                      "converged": st.n_converged, "iterations_per_s": it / dt, "ms_per_batched_solve": 1e3 * dt / reps,
                      "kernel_ms": kms / reps, "alg_GBps": ab / (kms * 1e-3) / 1e9}))


def mpc(name, prob, x0, u_guess, resolves, replan, move=None, **kw):
    B = len(x0)
    s = make(prob, B, **kw)
    def run():
        x_nom = np.array(prob["x_nom"], float)
        s.Reset(); s.SetTargetState(x_nom)
        s.SetInitialState(x0); s.SetInitialGuess(u_guess); s._push_problem()
        it = 0; kms = 0; ab = 0
        st = s.solve_resident(); it += st.total_iters; kms += st.kernel_ms; ab += st.algorithmic_bytes
        step = None
        if move is not None:
            step = np.zeros(s.n); step[move[0]] = move[1]
        st = s.MPCRun(resolves, replan, target_step=step)       # the whole receding-horizon loop on the device
        it += st.total_iters; kms += st.kernel_ms; ab += st.algorithmic_bytes
        return it, kms, ab
    run()
    t0 = time.perf_counter(); it, kms, ab = run(); dt = time.perf_counter() - t0
    print(json.dumps({"config": name, "B": B, "solves": resolves + 1, "iters_total": it, "iterations_per_s": it / dt,
                      "ms_per_batched_solve": 1e3 * dt / (resolves + 1), "ms_total": 1e3 * dt, "kernel_ms_total": kms,
                      "alg_GBps": ab / (kms * 1e-3) / 1e9}))


p = W.pendulum_problem()
single("C1 pendulum single", p, np.zeros((1, 2)), np.zeros((1, 199)))
single("C2 pendulum B=1024", p, W.pendulum_batch_x0(1024), np.zeros((1, 199)))
a = W.acrobot_problem()
mpc("C3 acrobot MPC B=512 x 51 solves", a, W.acrobot_batch_x0(512), np.zeros((1, a["N"] - 1)), 50, 2)
c = W.cartpole_wall_problem()
single("C4 cartpole_wall B=256 FD", c, W.cartpole_wall_batch_x0(256), np.zeros((1, 199)), reps=3)
q = W.synth36_problem()
mpc("C5 synth36 MPC B=64 x 101 solves", q, W.synth36_batch_x0(64), W.synth36_u_guess(q["N"]), 100, 4,
    move=(0, W.SYNTH_TARGET_VEL * q["dt"] * 4))
pq = W.planar_quad_problem()
mpc("C5q planar quadruped MPC B=64 x 101 solves", pq, W.planar_quad_batch_x0(64), W.planar_quad_u_guess(pq["N"]), 100, 4,
    move=(0, W.QUAD_TARGET_VEL * pq["dt"] * 4))
q3 = W.quad3d_problem()
mpc("C5q3d 3-D quadruped MPC B=64 x 101 solves", q3, W.quad3d_batch_x0(64), W.quad3d_u_guess(q3["N"]), 100, 4,
    move=(4, W.QUAD3D_TARGET_VEL * q3["dt"] * 4))
a27 = W.arm27_problem()
mpc("C6 arm + ball MPC B=64 x 21 solves", a27, W.arm27_batch_x0(64), W.arm27_u_guess(a27["N"]), 20, 5)
if os.environ.get("MI_RUN_SHARD") == "1":     # (same grid as C5 - 8 problems x 8 workgroups: kept out of the counter passes)
    mpc("C5/8GPU shard: synth36 MPC B=8 x 101 solves", q, W.synth36_batch_x0(64)[:8], W.synth36_u_guess(q["N"]), 100, 4,
        move=(0, W.SYNTH_TARGET_VEL * q["dt"] * 4))
a27c = W.arm27c_problem()
mpc("C6b coupled arm + ball MPC B=64 x 21 solves", a27c, W.arm27_batch_x0(64), W.arm27c_u_guess(a27c["N"]), 20, 5)
