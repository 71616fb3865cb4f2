"""Counters of the C ABI (crn_sense_get_stats / crn_sense_set_timing / crn_ingest_get_stats): what an operator reads instead of the
reference's printf lines.  The ring's counters are also asserted without a GPU in tests/harness/ring_unit.cpp."""
import ctypes as C

import numpy as np
import pytest

import crnsense as cs


def test_stats_argument_errors(built):
    L = cs.lib()
    st = cs.SenseStats()
    assert L.crn_sense_get_stats(None, C.byref(st)) == cs.CRN_ERR_ARG
    assert L.crn_sense_set_timing(None, 1) == cs.CRN_ERR_ARG
    assert L.crn_ingest_get_stats(None, None) == cs.CRN_ERR_ARG


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["energy4096", "welch4096", "ref512_L364"])
def test_sense_counters_and_kernel_time(built, mode):
    """launches / epochs / samples count what was launched (Welch epochs share their overlap: every sample once), and with timing on
    the per-launch durations agree with events recorded around the same launches on the same stream."""
    import torch
    dev = torch.device("cuda", 0)
    cfg = {"energy4096": lambda: cs.cfg_energy_scaled(4096, 4.0), "welch4096": lambda: cs.cfg_welch(4096, 8, 64),
           "ref512_L364": cs.cfg_reference}[mode]()
    L = 364 if mode == "ref512_L364" else cfg.fft_len
    E = 4096
    need = cs.samples_needed(cfg, E, L)
    iq = (torch.randn(need * 2, device=dev) * 1e-3).contiguous()
    outs = {"features": torch.empty(E, cfg.n_bands, device=dev).data_ptr(),
            "ann_out": torch.empty(E, 3, dtype=torch.float64, device=dev).data_ptr(),
            "decision": torch.empty(E, dtype=torch.int32, device=dev).data_ptr(),
            "occupancy": torch.empty(E, cfg.n_bands, dtype=torch.uint8, device=dev).data_ptr(), "spectrum": 0}
    s = cs.Sensor(cfg)
    stream = torch.cuda.current_stream().cuda_stream
    s.run_device(iq.data_ptr(), E, L, outs, stream=stream)            # untimed
    st = s.stats()
    assert (st["launches"], st["epochs"], st["samples"], st["timed_launches"]) == (1, E, need, 0)
    s.set_timing(True)
    n = 40                                                             # more than the 16 event pairs: slots are reused
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record()
        s.run_device(iq.data_ptr(), E, L, outs, stream=stream)
        b.record()
    torch.cuda.synchronize()
    st = s.stats()
    assert (st["launches"], st["epochs"], st["samples"], st["timed_launches"]) == (1 + n, (1 + n) * E, (1 + n) * need, n)
    outer = np.array([a.elapsed_time(b) for a, b in ev])
    assert st["kernel_ms"] <= outer.sum() * 1.02 and st["kernel_ms"] > 0.5 * outer.sum()    # inner events: a little less than the outer ones
    assert st["kernel_ms_min"] <= st["kernel_ms_last"] <= st["kernel_ms_max"] <= outer.max() * 1.05 + 0.02
    s.set_timing(False)
    s.run_device(iq.data_ptr(), E // 2, L, outs, stream=stream)
    torch.cuda.synchronize()
    st2 = s.stats()
    assert st2["launches"] == n + 2 and st2["timed_launches"] == n and st2["epochs"] == (1 + n) * E + E // 2
    s.close()


@pytest.mark.gpu
def test_ingest_counters(built):
    cfg = cs.cfg_reference()
    s = cs.Sensor(cfg)
    s.set_timing(True)
    ring = cs.Ingest(s, 3, 364, 4)
    rng = np.random.default_rng(5)
    got = 0
    for e in range(8):
        for p in range(10):
            for st in range(3):
                ring.push(st, (rng.normal(0, 1e-3, 364 * 2)).astype(np.float32))
        got += len(ring.poll())
    ring.drain()
    got += len(ring.poll())
    st = ring.stats()
    assert got == 24 and st["packets"] == 240 and st["epochs_launched"] == 24 and st["epochs_ready"] == 24 and st["epochs_polled"] == 24
    assert st["batches"] >= 6 and st["batches_failed"] == 0 and 0 < st["latency_us_max"] < 1e6
    sst = s.stats()                                                     # the ring launches through the same handle
    # (the handle also counts the slots launched empty: epochs that were open when a batch left moved on to the other buffer)
    assert sst["launches"] == st["batches"] and 24 <= sst["epochs"] <= 24 + 3 * st["batches"] and sst["timed_launches"] == st["batches"]
    ring.close()
    s.close()


@pytest.mark.gpu
def test_engine_summary_line(built, tmp_path):
    """`-s 1` in the engine's ce_args: every launch timed, one summary line at release() — epochs, launches, the samples the launches
    covered (10 packets of 364 per epoch), kernel time per launch, hand-off-to-decision latency of the ring."""
    import re
    import signals
    from test_gpu_parity import _run_harness
    cfg = cs.cfg_reference()
    L, n_epochs = 364, 12
    iq, _ = signals.make_epochs(cfg, n_epochs, seed=77, L=L)
    out = _run_harness("engine_harness", ["IQ", str(L), "-g", "0", "-v", "0", "-s", "1"], tmp_path, iq)
    line = [ln for ln in out if ln.startswith("CE_Predictive_Node_GPU:")]
    assert len(line) == 1, out[-5:]
    m = re.match(r"CE_Predictive_Node_GPU: epochs (\d+)  launches (\d+)  samples (\d+)  packets refused (\d+)  kernel ([\d.]+) us mean "
                 r"\(([\d.]+) \.\. ([\d.]+)\)  hand-off to decision ([\d.]+) us mean, ([\d.]+) us max", line[0])
    assert m, line[0]
    epochs, launches, samples, refused = (int(m.group(i)) for i in range(1, 5))
    kmean, kmin, kmax, lmean, lmax = (float(m.group(i)) for i in range(5, 10))
    # + the constructor's warm-up launch (crn_sense_reserve_host: one epoch of full-length frames, so that the kernel is loaded)
    assert (epochs, launches, samples, refused) == (n_epochs, n_epochs + 1, n_epochs * 10 * L + 10 * 512, 0)
    assert 1.0 < kmin <= kmean <= kmax < 5000.0 and kmean <= lmean <= lmax < 1e6
