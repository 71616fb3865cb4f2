"""Multi-GPU layout of the sensing path: shard streams, gather occupancy.

Every (stream, epoch) is independent — the only state that crosses frames is the K-frame mean
inside one epoch of one stream (reference: fft_avg[], CE_Predictive_Node.hpp:51) — so the path
shards by stream with no data-path collective.  The one exchange is the occupancy vector: every
node's engine needs the full picture to pick a free channel, so each rank contributes its
[epochs, n_bands] uint8 block to an all-gather.  Messages are a few KiB: latency-bound, one
collective per batch.

On the GPU the exchange is the C ABI's (`crn_comm_*` in include/crn_sense.h: RCCL over xGMI, side
stream, double-buffered slots) — `DeviceOccupancyExchange` below only carries the RCCL unique id
from rank 0 to the others over the launcher's control group (gloo).  `OccupancyExchange` is the same
slot logic on CPU tensors over gloo, for the world-size-2 layout tests that run without a GPU.
"""
import torch.distributed as dist


def shard(n_total, rank, world):
    """Contiguous block [lo, hi) of `n_total` streams owned by `rank` (remainder to the low ranks)."""
    base, rem = divmod(n_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_occupancy(occ_local, occ_all=None):
    """All-gather equal-sized per-rank occupancy blocks (CPU tensors, gloo); returns [world * epochs, n_bands]."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    if world == 1:
        return occ_local
    if occ_all is None:
        occ_all = occ_local.new_empty((world * occ_local.shape[0],) + tuple(occ_local.shape[1:]))
    dist.all_gather_into_tensor(occ_all, occ_local)
    return occ_all


class OccupancyExchange:
    """CPU twin (gloo) of the slot logic of crn_comm_*: step i writes slot i % depth (`local(i)`),
    `exchange(i)` gathers it; used by tests/test_sharding_gloo.py."""

    def __init__(self, epochs, n_bands, device="cpu", depth=2):
        import torch
        if torch.device(device).type != "cpu":
            raise ValueError("on the GPU use DeviceOccupancyExchange (the C ABI's crn_comm_*)")
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.depth = depth
        self.local_bufs = [torch.empty(epochs, n_bands, dtype=torch.uint8) for _ in range(depth)]
        self.all_bufs = [torch.empty(self.world * epochs, n_bands, dtype=torch.uint8) for _ in range(depth)]

    def local(self, i):
        return self.local_bufs[i % self.depth]

    def exchange(self, i):
        s = i % self.depth
        if not dist.is_initialized():
            self.all_bufs[s].copy_(self.local_bufs[s])
        else:
            dist.all_gather_into_tensor(self.all_bufs[s], self.local_bufs[s])
        return self.all_bufs[s]

    def gathered(self, i):
        return self.all_bufs[i % self.depth]

    def finish(self):
        pass


def make_device_exchange(epochs, n_bands, device_index, rank, world, depth=2):
    """The C ABI's exchange (crn_comm_*: RCCL loaded by libcrnsense itself) — the one backend this path has.  Every rank first
    draws a throw-away unique id, which is not collective, and the ranks agree over the control group whether all of them could:
    a rank that cannot load RCCL must not leave the others waiting inside ncclCommInitRank.  If any rank cannot, every rank
    exits non-zero (the launcher then takes the job down); nothing falls back to another collective.
    Returns (exchange, description)."""
    import sys
    import torch
    import crnsense as cs
    ok, why = 1, ""
    try:
        cs.comm_unique_id()
    except Exception as e:   # noqa: BLE001 — any failure means "RCCL cannot be used through the C ABI on this rank"
        ok, why = 0, str(e)
    all_ok = ok
    if world > 1:
        flag = torch.tensor([ok], dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        all_ok = int(flag.item())
    if not all_ok:
        msg = (f"RCCL is not usable through crn_comm_* here ({why[:300]})" if not ok else "another rank cannot load RCCL through crn_comm_*")
        print(f"sharding: rank {rank}: " + msg + ": stopping (exit 3)", file=sys.stderr)
        stop = SystemExit(3)
        stop.crn_reason = msg       # (bench.py's failure report quotes it)
        raise stop
    return (DeviceOccupancyExchange(epochs, n_bands, device_index, rank, world, depth),
            "RCCL all-gather of occupancy (crn_comm_*, side stream)")


class DeviceOccupancyExchange:
    """The occupancy exchange on the GPU: libcrnsense's crn_comm_* (RCCL all-gather on a side stream,
    `depth` slots).  rank / world come from the launcher (torch.distributed.run); the RCCL unique id is
    created on rank 0 through the C ABI and broadcast over the control group (any backend: gloo here)."""

    def __init__(self, epochs, n_bands, device_index, rank, world, depth=2):
        import crnsense as cs
        self.cs = cs
        self.epochs, self.n_bands, self.rank, self.world, self.depth = epochs, n_bands, rank, world, depth
        box = [cs.comm_unique_id() if rank == 0 else None]
        if world > 1:
            dist.broadcast_object_list(box, src=0)
        self.comm = cs.Comm(device_index, rank, world, box[0], epochs * n_bands, depth)

    def local_ptr(self, i, stream):
        """Device address step i's kernel writes its occupancy block to."""
        return self.comm.local(i, stream)

    def exchange(self, i, stream):
        self.comm.allgather(i, stream)

    def finish(self, stream):
        self.comm.finish(stream)

    def gathered_host(self, i):
        """Step i's gathered vector as a numpy [world * epochs, n_bands] array (after a synchronise)."""
        import numpy as np
        raw = self.cs.device_to_host(self.comm.gathered(i), self.world * self.epochs * self.n_bands)
        return np.frombuffer(raw, dtype=np.uint8).reshape(self.world * self.epochs, self.n_bands)

    def local_host(self, i):
        import numpy as np
        raw = self.cs.device_to_host(self.comm.local_addr(i), self.epochs * self.n_bands)   # an accessor: no stream waits, slot state untouched
        return np.frombuffer(raw, dtype=np.uint8).reshape(self.epochs, self.n_bands)

    def info(self):
        """crn_comm_info: what RCCL itself reports for this rank's communicator."""
        return self.comm.info()

    def close(self):
        self.comm.close()
