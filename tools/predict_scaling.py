"""What `bench.py --gpus N` should report per config on an N-GPU node, measured on ONE GPU (no 8-GPU node is available to the
builder; the driver's SCALE record is to be compared with this):  python tools/predict_scaling.py <tag>
For every config of bench.all_configs and N in {1, 2, 4, 8}: the N contiguous shards (dist.shard_range - exactly what the ranks
would own) are run one after the other on this GPU through bench.run_config; the predicted whole-node value is
(iterations of all shards) / (the SLOWEST shard's time) - ranks run concurrently, the fences wait for the slowest, and the path has
no data-path collective.  Writes gpurun_out/<tag>_predicted_scaling.json."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
torch.cuda.init()               # (before the library touches the device: torch's lazy initialization fails after it)
import bench
from drake_ddp_amd import workloads as W
from drake_ddp_amd.dist import shard_range

tag = sys.argv[1] if len(sys.argv) > 1 else "rXX"


class One:                       # bench.Ranks for a single process
    world, rank = 1, 0
    def fence(self):
        import torch
        torch.cuda.synchronize()
    def reduce(self, values, op):
        return [float(v) for v in values]


def configs():
    a, c, q = W.acrobot_problem(), W.cartpole_wall_problem(), W.synth36_problem()
    pq, q3, a27 = W.planar_quad_problem(), W.quad3d_problem(), W.arm27_problem()
    p = W.pendulum_problem()
    yield "C2", p, W.pendulum_batch_x0(1024), np.zeros((1, p["N"] - 1)), 10, None, True      # weak: every rank its own 1024
    yield "C3", a, W.acrobot_batch_x0(512), np.zeros((1, a["N"] - 1)), 3, (50, 2, None), False
    yield "C4", c, W.cartpole_wall_batch_x0(256), np.zeros((1, c["N"] - 1)), 3, None, False
    yield "C5", q, W.synth36_batch_x0(64), W.synth36_u_guess(q["N"]), 2, (100, 4, (0, W.SYNTH_TARGET_VEL * q["dt"] * 4)), False
    yield "C5q", pq, W.planar_quad_batch_x0(64), W.planar_quad_u_guess(pq["N"]), 2, (100, 4, (0, W.QUAD_TARGET_VEL * pq["dt"] * 4)), False
    yield "C5q3d", q3, W.quad3d_batch_x0(64), W.quad3d_u_guess(q3["N"]), 2, (100, 4, (4, W.QUAD3D_TARGET_VEL * q3["dt"] * 4)), False
    yield "C6", a27, W.arm27_batch_x0(64), W.arm27_u_guess(a27["N"]), 2, (20, 5, None), False
    a27c = W.arm27c_problem()
    yield "C6b", a27c, W.arm27_batch_x0(64), W.arm27c_u_guess(a27c["N"]), 2, (20, 5, None), False


rk, out = One(), {}
for name, prob, x0, ug, reps, mpc, weak in configs():
    row = {}
    for N in (1, 2, 4, 8):
        its, times = 0.0, []
        for r in range(1 if weak else N):
            lo, hi = (0, len(x0)) if weak else shard_range(len(x0), r, N)
            if hi == lo:
                continue
            rec = bench.run_config(rk, 0, name, prob, x0[lo:hi], ug, reps, mpc=mpc)
            its += rec["iterations"]; times.append(rec["iterations"] / rec["iterations_per_s"])
        if weak:
            its, times = its * N, times
        row[str(N)] = {"iterations_per_s": its / max(times), "slowest_shard_s": max(times), "fastest_shard_s": min(times)}
    base = row["1"]["iterations_per_s"]
    for N in row:
        row[N]["x_one_gpu"] = row[N]["iterations_per_s"] / base
    out[name] = row
    print(name, {N: (round(v["iterations_per_s"]), round(v["x_one_gpu"], 2)) for N, v in row.items()}, flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump({"method": __doc__, "configs": out}, open(os.path.join(ROOT, "gpurun_out", tag + "_predicted_scaling.json"), "w"), indent=1)
