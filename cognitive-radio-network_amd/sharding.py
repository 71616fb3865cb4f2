"""Multi-GPU layout of the sensing path: shard streams, gather occupancy.

Every (stream, epoch) is independent — the only state that crosses frames is the K-frame mean
inside one epoch of one stream (reference: fft_avg[], CE_Predictive_Node.hpp:51) — so the path
shards by stream with no data-path collective.  The one exchange is the occupancy vector: every
node's engine needs the full picture to pick a free channel, so each rank contributes its
[epochs, n_bands] uint8 block to an all-gather (RCCL over xGMI on the GPU box; gloo in CPU tests).
Messages are a few KiB: latency-bound, one collective per batch.
"""
import torch.distributed as dist


def shard(n_total, rank, world):
    """Contiguous block [lo, hi) of `n_total` streams owned by `rank` (remainder to the low ranks)."""
    base, rem = divmod(n_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_occupancy(occ_local, occ_all=None, force=False):
    """All-gather equal-sized per-rank occupancy blocks; returns the [world * epochs, n_bands] tensor.
    `force` runs the collective even in a group of one rank (single-GPU dry run of the N>1 path)."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    if world == 1 and not (force and dist.is_initialized()):
        return occ_local
    if occ_all is None:
        occ_all = occ_local.new_empty((world * occ_local.shape[0],) + tuple(occ_local.shape[1:]))
    dist.all_gather_into_tensor(occ_all, occ_local)
    return occ_all


class OccupancyExchange:
    """Double-buffered occupancy all-gather that overlaps the next launch.

    Step i writes its occupancy block into slot i % depth (`local(i)`), then `exchange(i)` queues
    the all-gather of that slot on a side stream behind an event, so the sensing kernel of step
    i + 1 starts without waiting for the collective (a few KiB per rank over xGMI: latency, not
    bandwidth).  A slot is handed out again only after its previous gather has finished
    (`local()` makes the launch stream wait for it).  On CPU tensors (gloo tests) the same calls
    run synchronously.
    """

    def __init__(self, epochs, n_bands, device, depth=2):
        import torch
        self._torch = torch
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.depth = depth
        self.cuda = torch.device(device).type == "cuda"
        self.local_bufs = [torch.empty(epochs, n_bands, dtype=torch.uint8, device=device) for _ in range(depth)]
        self.all_bufs = [torch.empty(self.world * epochs, n_bands, dtype=torch.uint8, device=device)
                         for _ in range(depth)]
        if self.cuda:
            self.side = torch.cuda.Stream(device=device)
            self.ready = [torch.cuda.Event() for _ in range(depth)]   # launch stream: block written
            self.done = [torch.cuda.Event() for _ in range(depth)]    # side stream: gather finished
            self.pending = [False] * depth

    def local(self, i):
        """Block step i's kernel writes (waits, on the launch stream, for the slot's last gather)."""
        s = i % self.depth
        if self.cuda and self.pending[s]:
            self._torch.cuda.current_stream().wait_event(self.done[s])
            self.pending[s] = False
        return self.local_bufs[s]

    def exchange(self, i):
        """Queue the all-gather of step i's block; returns the (eventually) gathered tensor."""
        s = i % self.depth
        if not dist.is_initialized():
            self.all_bufs[s].copy_(self.local_bufs[s])
            return self.all_bufs[s]
        if not self.cuda:
            dist.all_gather_into_tensor(self.all_bufs[s], self.local_bufs[s])
            return self.all_bufs[s]
        torch = self._torch
        self.ready[s].record(torch.cuda.current_stream())
        with torch.cuda.stream(self.side):
            self.side.wait_event(self.ready[s])
            dist.all_gather_into_tensor(self.all_bufs[s], self.local_bufs[s])
            self.done[s].record(self.side)
        self.pending[s] = True
        return self.all_bufs[s]

    def gathered(self, i):
        return self.all_bufs[i % self.depth]

    def finish(self):
        """Make the launch stream wait for every queued gather (call before the final synchronize)."""
        if self.cuda:
            for s in range(self.depth):
                if self.pending[s]:
                    self._torch.cuda.current_stream().wait_event(self.done[s])
                    self.pending[s] = False
