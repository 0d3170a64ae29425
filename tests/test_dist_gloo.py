"""N>1 path on CPU: world_size-2 gloo process group.  Each rank owns a contiguous
shard (drake_ddp_amd.dist.shard_range), produces per-problem costs for its shard (the
oracle stands in for the device here — there is no GPU in this container), and the
SAME collective helpers the product uses must return the global best on every rank."""
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    from drake_ddp_amd import workloads as W
    from drake_ddp_amd.dist import shard_range, allreduce_min, best_of_all_ranks, allreduce_sum
    from common import make_oracle
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    prob = W.pendulum_problem()
    B = 5
    x0 = W.pendulum_batch_x0(B, seed=11)
    lo, hi = shard_range(B, rank, world)
    costs, iters = [], 0
    for b in range(lo, hi):
        o = make_oracle(prob, jacobian="fd")
        o.set_problem(x0[b], prob["x_nom"], prob["Q"], prob["R"], prob["Qf"], np.zeros((1, prob["N"] - 1)))
        _, _, L, hist = o.solve()
        costs.append(L)
        iters += len(hist)
    li = int(np.argmin(costs))
    best = allreduce_min(costs[li])
    cost, gidx, owner = best_of_all_ranks(costs[li], li, lo)
    tot = allreduce_sum([iters, hi - lo])
    out.put((rank, lo, hi, costs, best, cost, gidx, owner, tot.tolist()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_world2_shard_and_best_cost_reduction():
    world = 2
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(out.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1:3] == (0, 3) and res[1][1:3] == (3, 5)            # contiguous, sizes differ by <= 1
    all_costs = res[0][3] + res[1][3]
    gbest = min(all_costs)
    for r in res:
        assert r[4] == gbest and r[5] == gbest                        # every rank sees the global min
        assert r[6] == int(np.argmin(all_costs))                      # global problem index of the winner
        assert r[7] == (0 if r[6] < 3 else 1)
        assert r[8][1] == 5                                           # summed shard sizes = global batch
    assert res[0][8] == res[1][8]


def _worker_async(rank, world, port, out):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from drake_ddp_amd.dist import allreduce_min_async
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from drake_ddp_amd.dist import allreduce_min_vec_async
    hs = [allreduce_min_async(10.0 * (rank + 1) + k) for k in range(3)]     # several in flight
    # the bench's grouped form: one element-wise reduction of a group's best costs
    hv = allreduce_min_vec_async([5.0 - rank, 7.0 + rank, 1.0, 2.0 * rank])
    out.put((rank, [h.wait() for h in hs], [float(v) for v in hv.wait()]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_world2_async_min_overlaps():
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker_async, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(out.get(timeout=100) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == [10.0, 11.0, 12.0] and res[1][1] == [10.0, 11.0, 12.0]
    assert res[0][2] == [4.0, 7.0, 1.0, 0.0] and res[1][2] == [4.0, 7.0, 1.0, 0.0]
