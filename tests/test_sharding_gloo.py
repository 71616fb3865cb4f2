"""World-size-2 run of the multi-GPU layout on CPU (gloo): stream sharding + occupancy all-gather.
The per-rank sensing itself needs a GPU, so here each rank's block is produced by the CPU oracle
(test infrastructure) — what is under test is cognitive-radio-network_amd/sharding.py, the code
bench.py runs between launches on the GPU box."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import crnsense as cs
import oracle_py as orc
import signals
from sharding import gather_occupancy, shard


def test_shard_covers_everything_once():
    for n in (1, 7, 8, 256, 1000):
        for w in (1, 2, 3, 8):
            blocks = [shard(n, r, w) for r in range(w)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in blocks]
            assert max(sizes) - min(sizes) <= 1


def _make_cfg(kind):
    if kind == "scan64":  # cfg4's shard unit: Welch PSD, 64 channels per stream
        cfg = cs.cfg_welch(1024, 4, 64)
        for b in range(64):
            cfg.thresh[b] = 1e-2
        return cfg
    return cs.cfg_energy_scaled(1024, 4.0)


def _worker(rank, world, port, n_epochs, q, kind="energy"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = _make_cfg(kind)
    iq, _ = signals.make_epochs(cfg, n_epochs, seed=31337)          # same on every rank
    lo, hi = shard(n_epochs, rank, world)
    spe = cs.samples_per_epoch(cfg)
    mine = orc.run(cfg, iq[lo * spe * 2:(hi * spe + cs.samples_needed(cfg, 0)) * 2], hi - lo)  # this rank's streams only
    occ_all = gather_occupancy(torch.from_numpy(mine["occupancy"]))
    if rank == 0:
        q.put(occ_all.numpy().copy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("kind,n_bands", [("energy", 4), ("scan64", 64)])
def test_two_rank_gather_equals_single_process(built, kind, n_bands):
    world, n_epochs = 2, 12
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_epochs, q, kind)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    cfg = _make_cfg(kind)
    iq, picks = signals.make_epochs(cfg, n_epochs, seed=31337)
    want = orc.run(cfg, iq, n_epochs)["occupancy"]
    assert np.array_equal(got, want)
    assert got.shape == (n_epochs, n_bands)
    assert got.any() and not got.all()


def _exchange_worker(rank, world, port, q):
    from sharding import OccupancyExchange
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    epochs, n_bands = 5, 4
    ex = OccupancyExchange(epochs, n_bands, "cpu")
    out = []
    for i in range(5):  # more steps than slots: every slot is reused
        blk = ex.local(i)
        blk.copy_(torch.full((epochs, n_bands), 10 * i + rank, dtype=torch.uint8))
        out.append(ex.exchange(i).clone())
    ex.finish()
    if rank == 1:
        q.put([o.numpy() for o in out])
    dist.barrier()
    dist.destroy_process_group()


def test_overlapped_exchange_two_ranks():
    """sharding.OccupancyExchange (what bench.py runs between launches at N > 1): slot reuse and
    rank order of the gathered vector."""
    world = 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_exchange_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for i, g in enumerate(got):
        assert g.shape == (10, 4)
        assert (g[:5] == 10 * i).all() and (g[5:] == 10 * i + 1).all()


def test_exchange_without_group_is_a_copy():
    from sharding import OccupancyExchange
    ex = OccupancyExchange(3, 2, "cpu")
    ex.local(0).fill_(7)
    assert (ex.exchange(0) == 7).all() and ex.gathered(0).shape == (3, 2)
