"""crn_sense_run_device inside a hipGraph: a launch only enqueues (no allocation, no synchronisation, no host read of device memory), so
a launch-bound loop of small batches — a node sweeping many radios' separate buffers — can be captured once and replayed.  Checked:
the replayed graph writes exactly what the direct launches write, for fresh input each replay; the per-launch cost of both forms is
printed (and kept under $CRN_EVIDENCE_DIR)."""
import os
import time

import numpy as np
import pytest

import crnsense as cs
import oracle_py as orc
import signals

pytestmark = pytest.mark.gpu


def test_small_launches_captured_in_a_hip_graph(built):
    import torch
    dev = torch.device("cuda", 0)
    cfg = cs.cfg_reference()
    n_buf, E = 64, 8                                     # 64 separate buffers of 8 epochs each: one launch per buffer
    spe = cs.samples_per_epoch(cfg)
    host = [signals.make_epochs(cfg, E, seed=900 + i)[0] for i in range(n_buf)]
    bufs = [torch.from_numpy(h).to(dev) for h in host]
    dec = [torch.zeros(E, dtype=torch.int32, device=dev) for _ in range(n_buf)]
    feat = [torch.zeros(E, 4, dtype=torch.float32, device=dev) for _ in range(n_buf)]
    s = cs.Sensor(cfg)
    side = torch.cuda.Stream(device=dev)

    def sweep(stream):
        for i in range(n_buf):
            s.run_device(bufs[i].data_ptr(), E, 512, {"features": feat[i].data_ptr(), "ann_out": 0, "decision": dec[i].data_ptr(),
                                                      "occupancy": 0, "spectrum": 0}, stream=stream)
    sweep(side.cuda_stream)                              # direct, once: loads the code object, sets the kernel's attributes
    side.synchronize()
    direct = [(d.clone(), f.clone()) for d, f in zip(dec, feat)]
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        sweep(torch.cuda.current_stream().cuda_stream)   # captured: nothing runs yet
    for d, f in zip(dec, feat):
        d.zero_()
        f.zero_()
    g.replay()
    torch.cuda.synchronize()
    for i in range(n_buf):
        assert torch.equal(dec[i], direct[i][0]) and torch.equal(feat[i], direct[i][1]), i
    want = orc.run(cfg, host[5], E)
    assert np.array_equal(dec[5].cpu().numpy(), want["decision"])
    # fresh input, same graph: the buffers' CONTENTS are read at replay time
    new = signals.make_epochs(cfg, E, seed=4321)[0]
    bufs[7].copy_(torch.from_numpy(new).to(dev))
    g.replay()
    torch.cuda.synchronize()
    assert np.array_equal(dec[7].cpu().numpy(), orc.run(cfg, new, E)["decision"])
    # cost per launch, both forms (host time to issue + device time to drain, 20 sweeps each)
    def timed(fn):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / (20 * n_buf) * 1e6
    us_direct = timed(lambda: sweep(side.cuda_stream))
    us_graph = timed(g.replay)
    line = (f"hipGraph: {n_buf} launches of {E} reference epochs each per sweep: {us_direct:.2f} us per launch issued directly, "
            f"{us_graph:.2f} us per launch replayed from a captured graph")
    print(line)
    out_dir = os.environ.get("CRN_EVIDENCE_DIR")
    if out_dir and os.path.isdir(out_dir):
        open(os.path.join(out_dir, "hipgraph_small_launches.txt"), "w").write(line + "\n")
    s.close()


def test_updates_are_refused_on_a_capturing_stream(built):
    """crn_sense_set_thresholds / _set_ann / _calibrate_thresholds stage their values in a reused pinned slot and mark it with an event:
    captured into a graph, the event would never complete for the host and every replay would upload whatever the slot holds by then.
    They return CRN_ERR_STATE on a stream that is capturing and enqueue nothing — the capture itself is unharmed: the launches around
    the refused update are in the graph and replay with the thresholds set OUTSIDE the capture."""
    import torch
    dev = torch.device("cuda", 0)
    cfg = cs.cfg_energy_scaled(1024, 4.0)
    E = 8
    iq, _ = signals.make_epochs(cfg, E, seed=77)
    buf = torch.from_numpy(iq).to(dev)
    feat = torch.zeros(E, 4, device=dev)
    occ = torch.zeros(E, 4, dtype=torch.uint8, device=dev)
    outs = {"features": feat.data_ptr(), "ann_out": 0, "decision": 0, "occupancy": occ.data_ptr(), "spectrum": 0}
    s = cs.Sensor(cfg)
    side = torch.cuda.Stream(device=dev)
    s.run_device(buf.data_ptr(), E, 1024, outs, stream=side.cuda_stream)
    side.synchronize()
    before = occ.clone()
    assert int(before.sum()) > 0                          # the fixture's driven channels read occupied against lambda = 4
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        st = torch.cuda.current_stream().cuda_stream
        with pytest.raises(cs.CrnError, match="captured"):
            s.set_thresholds([float("inf")] * 4, stream=st)
        w = np.zeros((5, 6)), np.zeros((6, 4))
        assert cs.lib().crn_sense_set_ann(s._h, w[0].ctypes.data, w[1].ctypes.data, 0.8, st) != 0    # (refused: capture, then: not an ANN handle)
        s.run_device(buf.data_ptr(), E, 1024, outs, stream=st)
    occ.zero_()
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(occ, before)                       # the refused update left the thresholds alone
    s.set_thresholds([float("inf")] * 4, stream=side.cuda_stream)   # outside the capture: ordered on the stream, seen by the replay
    side.synchronize()
    g.replay()
    torch.cuda.synchronize()
    assert int(occ.sum()) == 0
    s.close()
