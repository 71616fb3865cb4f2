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


def gather_occupancy(occ_local, occ_all=None):
    """All-gather equal-sized per-rank occupancy blocks; returns the [world * epochs, n_bands] tensor."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    if world == 1:
        return occ_local
    if occ_all is None:
        occ_all = occ_local.new_empty((world * occ_local.shape[0],) + tuple(occ_local.shape[1:]))
    dist.all_gather_into_tensor(occ_all, occ_local)
    return occ_all
