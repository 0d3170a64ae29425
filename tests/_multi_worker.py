"""One rank of tests/test_gpu_multi.py (a plain script: every rank is its own process on its own GPU).

    python _multi_worker.py <rank> <world> <exchange_dir> [same]

"same": every rank uses GPU 0 (the communicator's ranks share one device - what a one-GPU box can run).

Checks, on `world` GPUs of one node, the path's one collective in the library's own RCCL communicator
(mi_ilqr_comm_*, NativeComm) and the shard partitioning of a batched solve; writes `<exchange_dir>/ok.<rank>`
(JSON) when every assertion held."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world, xdir = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    same = len(sys.argv) > 4 and sys.argv[4] == "same"
    dev = 0 if same else rank
    from drake_ddp_amd.dist import NativeComm, shard_range
    from drake_ddp_amd import workloads as W
    from drake_ddp_amd.ilqr import BatchedIterativeLQR
    from drake_ddp_amd.models import ModelSystem

    def exchange(ident):                                   # the 128-byte communicator id travels through a file
        path = os.path.join(xdir, "comm_id")
        if ident is not None:
            with open(path + ".tmp", "wb") as f:
                f.write(ident)
            os.replace(path + ".tmp", path)
            return ident
        t0 = time.time()
        while not os.path.exists(path):
            if time.time() - t0 > 120:
                raise TimeoutError("rank 0 never published the communicator id")
            time.sleep(0.05)
        with open(path, "rb") as f:
            return f.read()

    comm = NativeComm(rank, world, dev, exchange)
    # ranks contribute DIFFERENT values: min, not sum / average / max (round-2 advisor: the op code was ncclAvg)
    v = np.array([10.0 * (rank + 1), -3.0 * (rank + 1), 100.0 - rank, float(rank == world - 1)])
    want = np.array([10.0, -3.0 * world, 100.0 - (world - 1), 0.0 if world > 1 else 1.0])
    got = comm.allreduce_min(v)
    assert np.array_equal(got, want), (rank, got, want)
    got = comm.start(v[:2]).wait()
    assert np.array_equal(got, want[:2]), (rank, got)

    # the sharded solve: each rank solves its contiguous shard of the 64-problem batch on its own GPU and the
    # all-reduce(min) of the shards' best costs equals the minimum over the whole batch solved on ONE GPU
    p = W.pendulum_problem()
    B = 64
    x0 = W.pendulum_batch_x0(1024)[:B]
    lo, hi = shard_range(B, rank, world)

    def solve(x0s, device):
        s = BatchedIterativeLQR(ModelSystem(p["model_id"], p["dt"]), p["N"], len(x0s), delta=p["delta"], beta=p["beta"],
                                gamma=p["gamma"], device=device)
        s.SetTargetState(p["x_nom"]); s.SetRunningCost(p["Q"], p["R"]); s.SetTerminalCost(p["Qf"])
        s.SetInitialState(x0s); s.SetInitialGuess(np.zeros((1, p["N"] - 1)))
        s.Solve()
        return s
    mine = solve(x0[lo:hi], dev)
    best = comm.allreduce_min([mine.stats.best_cost])[0]
    out = {"rank": rank, "best": float(best), "iters": int(mine.iterations.sum()), "lo": lo, "hi": hi}
    if rank == 0:
        whole = solve(x0, 0)
        assert best == whole.cost.min(), (best, whole.cost.min())
        assert np.array_equal(whole.cost[lo:hi], mine.cost)           # a shard is bitwise its slice of the whole batch
        out["whole_iters"] = int(whole.iterations.sum())
    with open(os.path.join(xdir, f"ok.{rank}"), "w") as f:
        json.dump(out, f)


if __name__ == "__main__":
    main()
