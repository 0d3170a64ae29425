"""Multi-GPU plumbing: one process per GPU, each with its own handle over a
contiguous shard of the batch.  Problems are independent, so the
data path needs NO collective; the only exchange is the best-cost reduction at
the end of a batched solve — one RCCL all-reduce(min) of 8 bytes (plus an
all-gather of 16-byte {cost,index} pairs when the winner's identity is wanted).
torch.distributed is used as plumbing only (backend "nccl" = RCCL on ROCm; "gloo"
in the CPU tests)."""
import numpy as np


def shard_range(B, rank, world):
    """Contiguous block of problems owned by `rank`; sizes differ by at most one."""
    base, rem = divmod(B, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def _dist():
    import torch.distributed as dist
    return dist if (dist.is_available() and dist.is_initialized()) else None


def _device(device_id):
    import torch
    dist = _dist()
    if dist is not None and dist.get_backend() == "nccl":
        return torch.device("cuda", device_id)
    return torch.device("cpu")


def allreduce_min(value, device_id=0):
    """min over ranks of a scalar; identity without a process group."""
    import torch
    dist = _dist()
    if dist is None:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=_device(device_id))
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return float(t.item())


class PendingMin:
    """Handle of an in-flight all-reduce(min): lets the collective overlap the next solve."""

    def __init__(self, tensor, work):
        self.tensor, self.work = tensor, work

    def wait(self):
        if self.work is not None:
            self.work.wait()
            self.work = None
        return float(self.tensor.item())


def allreduce_min_async(value, device_id=0):
    """Non-blocking variant of allreduce_min (RCCL runs it on its own stream)."""
    import torch
    dist = _dist()
    if dist is None:
        return PendingMin(torch.tensor([float(value)], dtype=torch.float64), None)
    t = torch.tensor([float(value)], dtype=torch.float64, device=_device(device_id))
    return PendingMin(t, dist.all_reduce(t, op=dist.ReduceOp.MIN, async_op=True))


class PendingMinVec:
    """Handle of an in-flight element-wise all-reduce(min) of several values."""

    def __init__(self, tensor, work):
        self.tensor, self.work = tensor, work

    def wait(self):
        if self.work is not None:
            self.work.wait()
            self.work = None
        return self.tensor.cpu().numpy()


def allreduce_min_vec_async(values, device_id=0):
    """Element-wise min over ranks of a short vector (the best costs of a group of batched solves):
    ONE collective for the group instead of one per solve."""
    import torch
    dist = _dist()
    v = [float(x) for x in values]
    if dist is None:
        return PendingMinVec(torch.tensor(v, dtype=torch.float64), None)
    t = torch.tensor(v, dtype=torch.float64, device=_device(device_id))
    return PendingMinVec(t, dist.all_reduce(t, op=dist.ReduceOp.MIN, async_op=True))


def best_of_all_ranks(local_best_cost, local_best_index, shard_lo, device_id=0):
    """(cost, global problem index, owning rank) of the best converged problem of the
    whole batch: all-gather of one {cost, global index} pair per rank + local argmin."""
    import torch
    dist = _dist()
    gidx = float(shard_lo + local_best_index) if local_best_index >= 0 else -1.0
    if dist is None:
        return float(local_best_cost), int(gidx), 0
    mine = torch.tensor([float(local_best_cost), gidx], dtype=torch.float64, device=_device(device_id))
    allp = [torch.empty_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(allp, mine)
    costs = np.array([float(p[0]) for p in allp])
    r = int(np.argmin(costs))
    return float(costs[r]), int(allp[r][1].item()), r


def allreduce_sum(values, device_id=0):
    """Element-wise sum over ranks of a small vector (iteration counters for the metric)."""
    import torch
    dist = _dist()
    v = torch.tensor(np.asarray(values, dtype=np.float64), dtype=torch.float64, device=_device(device_id))
    if dist is not None:
        dist.all_reduce(v, op=dist.ReduceOp.SUM)
    return v.cpu().numpy()


class NativeComm:
    """The same collective without torch in the data path: an RCCL communicator owned by libmi_ilqr.so
    (include/mi_ilqr.h: mi_ilqr_comm_*), the way a C caller of the library reduces its best costs.  The
    128-byte communicator id is created on rank 0 and shipped by whatever channel the launcher has
    (`exchange`: a callable bytes-or-None -> bytes; from_torch() uses the process group's object broadcast)."""

    def __init__(self, rank, world, device_id, exchange=None, ident=None):
        import ctypes as C
        from . import _capi
        self._capi, self._C = _capi, C
        self._lib = _capi.load()
        if ident is None:
            if world > 1 and exchange is None:
                raise ValueError("world > 1 needs an `exchange` callable to ship the communicator id")
            # (a failure on rank 0 still goes through the exchange - as an empty id - so that the other ranks, which are waiting in
            #  it, fail with this rank instead of hanging)
            ident = self.unique_id() if rank == 0 else None
            if world > 1:
                ident = exchange(ident)
        if not ident:
            raise RuntimeError("mi_ilqr_comm_unique_id failed on rank 0 (librccl not loadable?)")
        h = C.c_void_p()
        idbuf = C.create_string_buffer(ident, _capi.COMM_ID_BYTES)
        _capi.check(self._lib.mi_ilqr_comm_create(idbuf, int(rank), int(world), int(device_id), C.byref(h)), "mi_ilqr_comm_create")
        self._h = h
        self.rank, self.world = int(rank), int(world)
        self._pending = 0

    @staticmethod
    def unique_id():
        """A fresh 128-byte communicator id (rank 0 creates it and ships it), b"" when librccl cannot make one."""
        import ctypes as C
        from . import _capi
        buf = (C.c_char * _capi.COMM_ID_BYTES)()
        return bytes(buf.raw) if _capi.load().mi_ilqr_comm_unique_id(buf) == _capi.OK else b""

    @staticmethod
    def torch_exchange(ident):
        """rank 0's id to every rank of the initialized torch.distributed process group (any backend)."""
        box = [ident]
        _dist().broadcast_object_list(box, src=0)
        return box[0]

    @classmethod
    def from_torch(cls, device_id):
        """Ranks and id exchange taken from the initialized torch.distributed process group (any backend)."""
        dist = _dist()
        if dist is None:
            return cls(0, 1, device_id)
        return cls(dist.get_rank(), dist.get_world_size(), device_id, cls.torch_exchange)

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._lib.mi_ilqr_comm_destroy(h)

    def count(self):
        """(ranks, this rank) as the COMMUNICATOR reports them (ncclCommCount / ncclCommUserRank)."""
        C = self._C
        n, r = C.c_int32(), C.c_int32()
        self._capi.check(self._lib.mi_ilqr_comm_count(self._h, C.byref(n), C.byref(r)), "mi_ilqr_comm_count")
        return int(n.value), int(r.value)

    def allreduce_min(self, values):
        v = np.ascontiguousarray(values, dtype=np.float64).copy()
        self._capi.check(self._lib.mi_ilqr_allreduce_min(self._h, self._capi.ptr(v), int(v.size)), "mi_ilqr_allreduce_min")
        return v

    def start(self, values):
        """Enqueue the reduction on the communicator's own stream (overlaps the next solves); wait() completes it."""
        v = np.ascontiguousarray(values, dtype=np.float64)
        self._capi.check(self._lib.mi_ilqr_allreduce_min_start(self._h, self._capi.ptr(v), int(v.size)), "mi_ilqr_allreduce_min_start")
        self._pending = int(v.size)
        return self

    def wait(self):
        out = np.empty(self._pending, dtype=np.float64)
        if self._pending:
            self._capi.check(self._lib.mi_ilqr_allreduce_min_wait(self._h, self._capi.ptr(out), self._pending), "mi_ilqr_allreduce_min_wait")
            self._pending = 0
        return out
