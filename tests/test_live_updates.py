"""Replacing what the reference hard-codes on a LIVE handle (SURVEY.md §8(b) proposed ABI): the band plan (the five loops of
CE_Predictive_Node.cpp:173-191 -> crn_sense_set_bands), the network (the literals of .cpp:78-120 -> crn_sense_set_ann, and the
weights file of crn_cfg_save_ann / crn_cfg_load_ann) and the thresholds from host-side features (crn_noise_floor_host).  CPU part:
the configuration helpers and the file format; GPU part: a handle after the update equals a fresh handle created with the new
configuration, bit for bit, and both equal the oracle."""
import ctypes as C

import numpy as np
import pytest

import crnsense as cs
import oracle_py as orc
import signals


def test_reference_scaled_and_welch_scaled_plans(built):
    ref = cs.cfg_reference()
    for n in (512, 1024, 2048, 4096):
        c = cs.cfg_reference_scaled(n)
        assert (c.fft_len, c.hop, c.mode, c.decide, c.n_bands, c.n_segs, c.frames_per_epoch) == (n, n, cs.MODE_REF_MAG, cs.DECIDE_ANN, 4, 5, 10)
        for i in range(5):
            assert (c.segs[i].lo, c.segs[i].hi, c.segs[i].band) == (ref.segs[i].lo * n // 512, ref.segs[i].hi * n // 512, ref.segs[i].band)
        assert all(c.ann_w_ih[i][j] == ref.ann_w_ih[i][j] for i in range(5) for j in range(6))
        w = cs.cfg_welch_scaled(n, 8, 3.0)
        e = cs.cfg_energy_scaled(n, 3.0)
        assert (w.hop, w.window, w.frames_per_epoch, w.ref_band) == (n // 2, cs.WINDOW_HANN, 8, 0)
        assert all(w.thresh[b] == e.thresh[b] for b in range(1, 4)) and all(w.segs[i].lo == e.segs[i].lo for i in range(5))
    c = cs.Cfg()
    assert cs.lib().crn_cfg_reference_scaled(C.byref(c), 768) == cs.CRN_ERR_ARG
    assert cs.lib().crn_cfg_welch_scaled(C.byref(c), 1024, 0, 4.0) == cs.CRN_ERR_ARG


def test_weights_file_round_trip_and_errors(built, tmp_path):
    rng = np.random.default_rng(3)
    cfg = cs.cfg_reference()
    wih, who = rng.normal(0, 3, (5, 6)), rng.normal(0, 10, (6, 4))
    wih[0, 0] = 1.0 / 3.0                      # not representable in few digits: the file must carry all 17
    cs.set_ann_weights(cfg, wih, who, threshold=0.7)
    path = tmp_path / "w.txt"
    cs.save_ann(cfg, str(path))
    text = path.read_text()
    assert text.startswith("#") and len(text.split("\n")) >= 12
    back = cs.load_ann(cs.cfg_energy_scaled(1024), str(path))     # everything else in the cfg stays
    assert all(back.ann_w_ih[i][j] == wih[i, j] for i in range(5) for j in range(6))
    assert all(back.ann_w_ho[j][k] == who[j, k] for j in range(6) for k in range(4))
    assert back.ann_threshold == 0.7 and back.fft_len == 1024 and back.mode == cs.MODE_ENERGY
    # the threshold is optional; comments anywhere; anything else is an error, not a partial load
    nums = [ln for ln in text.split("\n") if ln and not ln.startswith("#")]
    (tmp_path / "w54.txt").write_text("# no threshold line\n" + "\n".join(nums[:-1]) + "\n# trailing comment\n")
    b2 = cs.load_ann(cs.cfg_reference(), str(tmp_path / "w54.txt"))
    assert b2.ann_threshold == 0.8 and b2.ann_w_ho[5][3] == who[5, 3]
    for bad in ("1 2 3\n", "\n".join(nums) + "\n4.0\n", "\n".join(nums[:-1]) + "\nnan\n", "\n".join(nums[:-2]) + "\nx1.0\n"):
        (tmp_path / "bad.txt").write_text(bad)
        keep = cs.cfg_reference()
        assert cs.lib().crn_cfg_load_ann(C.byref(keep), str(tmp_path / "bad.txt").encode()) == cs.CRN_ERR_ARG
        assert keep.ann_w_ih[1][1] == cs.cfg_reference().ann_w_ih[1][1]     # untouched
    assert cs.lib().crn_cfg_load_ann(C.byref(cfg), str(tmp_path / "missing.txt").encode()) == cs.CRN_ERR_ARG
    assert cs.lib().crn_cfg_save_ann(C.byref(cfg), b"/nonexistent-dir/w.txt") == cs.CRN_ERR_ARG


def test_build_info_states_the_toolchain(built):
    built_hip, runtime_hip, ok = cs.build_info()
    assert built_hip // 10000000 >= 7              # gfx950 needs ROCm 7
    if runtime_hip:                                # (a box without any HIP runtime reports 0 and `ok` False)
        assert ok == (runtime_hip // 10000000 == built_hip // 10000000 and runtime_hip >= 70000000)


@pytest.mark.gpu
@pytest.mark.parametrize("n", [512, 4096])
def test_set_bands_equals_a_fresh_handle(built, n):
    """The band plan replaced on a live handle — reference plan -> three other bands + a reference band elsewhere (which also
    leaves the registers the reference plan reaches: the pruned kernel must give way) -> 20 bands (the LDS form of the close) -> back
    — gives, launch after launch, what a handle created with that plan gives, bit for bit; and the oracle's features."""
    base = cs.cfg_energy_scaled(n, 4.0)
    n_epochs = 19
    iq, _ = signals.make_epochs(base, n_epochs, seed=n + 1)
    s = cs.Sensor(base)
    first = s.run_host(iq, n_epochs)
    assert "PASS3_ROWS" in s.kernel_info()["name"]
    sc = n // 512
    plans = [
        ([(40 * sc, 60 * sc, 0), (100 * sc, 140 * sc, 1), (141 * sc, 150 * sc, 1), (260 * sc, 300 * sc, 2), (400 * sc, 470 * sc, 3)], 4, [np.inf, 2.0, 3.0, 1.5]),
        ([(b * (n // 20), (b + 1) * (n // 20) - 3, b) for b in range(20)], 20, [1e-3] * 20),
    ]
    for segs, nb, thr in plans:
        s.set_bands(segs, nb, thr)
        got = s.run_host(iq, n_epochs)
        fresh_cfg = cs.cfg_energy_scaled(n, 4.0)
        fresh_cfg.n_bands, fresh_cfg.n_segs = nb, len(segs)
        if nb == 20:
            fresh_cfg.ref_band = -1
            s.close()                            # ref_band is not part of set_bands: a plan with another reference is another handle
            s = cs.Sensor(fresh_cfg_with(fresh_cfg, segs, thr))
            got = s.run_host(iq, n_epochs)
        fresh = cs.Sensor(fresh_cfg_with(fresh_cfg, segs, thr))
        assert "PASS3_ROWS" not in fresh.kernel_info()["name"]
        want = fresh.run_host(iq, n_epochs)
        fresh.close()
        for k in ("features", "occupancy", "decision"):
            assert np.array_equal(got[k], want[k]), (k, nb)
        ref = orc.run(fresh_cfg_with(fresh_cfg, segs, thr), iq, n_epochs)
        assert (np.abs(got["features"] - ref["features"]) / np.maximum(np.abs(ref["features"]), 1e-30)).max() < 1e-5
    if s.cfg.n_bands == 20:
        s.close()
        s = cs.Sensor(cs.cfg_energy_scaled(n, 4.0))
        s.set_bands(plans[0][0], 4, plans[0][2])
    # and back to the reference plan: the first launch's results again, the pruned kernel again
    ref_plan = cs.cfg_energy_scaled(n, 4.0)
    s.set_bands([(ref_plan.segs[i].lo, ref_plan.segs[i].hi, ref_plan.segs[i].band) for i in range(5)], 4, [ref_plan.thresh[b] for b in range(4)])
    again = s.run_host(iq, n_epochs)
    assert "PASS3_ROWS" in s.kernel_info()["name"]
    for k in ("features", "occupancy", "decision"):
        assert np.array_equal(again[k], first[k]), k
    # argument errors leave the handle as it was
    with pytest.raises(cs.CrnError):
        s.set_bands([(0, n + 1, 0)], 1, [1.0])
    with pytest.raises(cs.CrnError):
        s.set_bands([(0, 8, 0), (8, 16, 1)], 2, None)          # another number of bands needs its thresholds
    ring = cs.Ingest(s, 1, n, 1)                               # a ring sized its result buffers for 4 bands:
    with pytest.raises(cs.CrnError, match="ingest ring"):
        s.set_bands([(0, 8, 0), (8, 16, 1)], 2, [1.0, 1.0])    # ... the number of bands cannot change under it
    s.set_bands(plans[0][0], 4, plans[0][2])                   # the plan can
    ring.close()
    s.set_bands([(ref_plan.segs[i].lo, ref_plan.segs[i].hi, ref_plan.segs[i].band) for i in range(5)], 4, [ref_plan.thresh[b] for b in range(4)])
    assert np.array_equal(s.run_host(iq, n_epochs)["features"], first["features"])
    s.close()


def fresh_cfg_with(cfg, segs, thr):
    for i, (lo, hi, b) in enumerate(segs):
        cfg.segs[i].lo, cfg.segs[i].hi, cfg.segs[i].band = lo, hi, b
    cfg.n_segs = len(segs)
    for b, t in enumerate(thr):
        cfg.thresh[b] = t
    return cfg


@pytest.mark.gpu
def test_set_ann_is_ordered_on_the_stream(built):
    """launch / crn_sense_set_ann / launch on one stream without a synchronise between them: the first launch decides with the old
    network, the second with the new one — each equal to a fresh handle's and to the oracle's forward pass."""
    import torch
    dev = torch.device("cuda", 0)
    cfg = cs.cfg_reference()
    n_epochs = 64
    iq, picks = signals.make_epochs(cfg, n_epochs, seed=8)
    rng = np.random.default_rng(12)
    wih, who = rng.normal(0, 0.3, (5, 6)), rng.normal(0, 4.0, (6, 4))
    d_iq = torch.from_numpy(iq).to(dev)
    outs = []
    for _ in range(2):
        outs.append((torch.zeros(n_epochs, 3, dtype=torch.float64, device=dev), torch.zeros(n_epochs, dtype=torch.int32, device=dev)))
    s = cs.Sensor(cfg)
    st = torch.cuda.Stream(device=dev)
    ptr = lambda o: {"features": 0, "ann_out": o[0].data_ptr(), "decision": o[1].data_ptr(), "occupancy": 0, "spectrum": 0}   # noqa: E731
    s.run_device(d_iq.data_ptr(), n_epochs, 512, ptr(outs[0]), stream=st.cuda_stream)
    s.set_ann(wih, who, threshold=0.6, stream=st.cuda_stream)
    s.run_device(d_iq.data_ptr(), n_epochs, 512, ptr(outs[1]), stream=st.cuda_stream)
    st.synchronize()
    s.close()
    old = orc.run(cs.cfg_reference(), iq, n_epochs)
    new = orc.run(cs.set_ann_weights(cs.cfg_reference(), wih, who, threshold=0.6), iq, n_epochs)
    assert np.abs(outs[0][0].cpu().numpy() - old["ann_out"]).max() < 1e-6 and np.array_equal(outs[0][1].cpu().numpy(), picks)
    assert np.abs(outs[1][0].cpu().numpy() - new["ann_out"]).max() < 1e-6
    assert np.abs(new["ann_out"] - old["ann_out"]).max() > 0.1                      # (the update did change the answer)
    margin = np.abs(new["ann_out"] - 0.6).min(axis=1) > 1e-3
    assert np.array_equal(outs[1][1].cpu().numpy()[margin], new["decision"][margin])
    # not for handles that do not decide with the network; bad weights refused
    e = cs.Sensor(cs.cfg_energy_scaled(1024))
    with pytest.raises(cs.CrnError):
        e.set_ann(wih, who)
    e.close()
    s2 = cs.Sensor(cs.cfg_reference())
    bad = wih.copy()
    bad[2, 2] = np.nan
    with pytest.raises(cs.CrnError):
        s2.set_ann(bad, who)
    with pytest.raises(cs.CrnError):
        s2.set_ann(wih, who, threshold=1.5)
    s2.close()


@pytest.mark.gpu
def test_noise_floor_from_host_features(built):
    """crn_noise_floor_host (what the engine's scan mode calls at start-up) = crn_noise_floor_device on the uploaded matrix = numpy's
    lower median of the per-epoch lower medians."""
    import torch
    cfg = cs.cfg_welch(1024, 8, 64)
    rng = np.random.default_rng(2)
    f = rng.gamma(4.0, 1e-3, (37, 64)).astype(np.float32)
    f[::3, 10:14] *= 50.0
    s = cs.Sensor(cfg)
    got = s.noise_floor_host(f)
    want = np.sort(np.sort(f, axis=1)[:, 31])[18]
    assert got == want
    d = torch.from_numpy(f).cuda()
    assert s.noise_floor(d.data_ptr(), 37) == want
    with pytest.raises(cs.CrnError):
        s.noise_floor_host(f[:0])
    check = cs.lib().crn_sense_synchronize(s._h, None)
    assert check == 0
    s.close()


@pytest.mark.gpu
def test_band_plan_swaps_against_a_running_ring_on_the_gpu(built):
    """The hardware counterpart of tests/harness/api_race_unit.cpp: one thread pushes packets through an ingest ring (its launcher
    thread launches through the handle) while another swaps the band plan between two plans — same number of bands — and the
    thresholds, 300 times.  Every epoch carries one tone; under plan A its bin lies in band 1, under plan B in band 2: each result
    must be exactly plan A's or plan B's answer (the tone's energy in ONE of the two bands, the other at noise level), never a
    mix, never an error from the launcher thread — and the handle cannot be destroyed while the ring is attached."""
    import threading
    n, K, L = 1024, 10, 1024
    cfg = cs.cfg_energy_scaled(n, 4.0)
    plan_a = [(600, 620, 0), (8, 24, 1), (110, 170, 2), (378, 444, 3)]
    plan_b = [(600, 620, 0), (40, 72, 1), (8, 24, 2), (378, 444, 3)]
    thr = [cfg.thresh[b] for b in range(4)]
    s = cs.Sensor(cfg)
    s.set_bands(plan_a, 4, thr)
    ring = cs.Ingest(s, 2, L, 4)
    with pytest.raises(cs.CrnError, match="ingest ring"):
        cs.check(cs.lib().crn_sense_destroy(s._h), "crn_sense_destroy")
    rng = np.random.default_rng(77)
    t = np.arange(L)
    pkt = (rng.normal(0, 7e-4, L) + 1j * rng.normal(0, 7e-4, L) + 0.02 * np.exp(2j * np.pi * 16 * t / n)).astype(np.complex64).view(np.float32).copy()
    stop, results, errors = threading.Event(), [], []

    def pusher():
        try:
            while not stop.is_set():
                for st in range(2):
                    ring.push(st, pkt, block=True)
                results.extend((r.features[1], r.features[2]) for r in ring.poll(64))
        except Exception as e:   # noqa: BLE001 — reported by the main thread
            errors.append(e)
    th = threading.Thread(target=pusher)
    th.start()
    try:
        for it in range(300):
            s.set_bands(plan_b if it & 1 else plan_a, 4, None)
            s.set_thresholds([t0 * (1.0 + 1e-3 * (it % 7)) for t0 in thr])
    finally:
        stop.set()
        th.join()
    assert not errors, errors
    ring.drain()
    results.extend((r.features[1], r.features[2]) for r in ring.poll(4096))
    ring.close()
    s.close()
    f = np.array(results, np.float64)
    assert len(f) > 50
    tone = (0.02 * n) ** 2                      # |X|^2 of the tone's bin, K-frame mean
    in1 = np.abs(f[:, 0] / tone - 1) < 0.05     # plan A: the tone (bin 16) lies in band 1 = [8, 24)
    in2 = np.abs(f[:, 1] / tone - 1) < 0.05     # plan B: in band 2 = [8, 24)
    assert (in1 ^ in2).all(), "an epoch saw a mix of the two plans (or neither)"
    assert in1.any() and in2.any()
    assert (np.where(in1, f[:, 1], f[:, 0]) < 1e-3 * tone).all()      # the other band holds noise only


@pytest.mark.gpu
def test_ring_calibration_and_one_call_calibration_on_the_gpu(built):
    """crn_ingest_calibrate on hardware: the ring's launcher thread measures the noise floor over the first epochs that come back and
    sets the thresholds; marked epochs first, then unmarked ones carrying the estimate — which equals crn_noise_floor_host's on the
    same features and crn_sense_calibrate_thresholds' (the synchronous engine's one allocation-free call), and the thresholds in force
    afterwards are lambda x it (a tone far above them reads occupied, noise bands idle)."""
    n, K = 1024, 8
    cfg = cs.cfg_welch(n, K, 64)
    for b in range(64):
        cfg.thresh[b] = float("inf")
    s = cs.Sensor(cfg)
    ring = cs.Ingest(s, 1, n, 1)
    ring.calibrate(6, 4.0)
    with pytest.raises(cs.CrnError):
        ring.calibrate(6, 4.0)                   # one at a time
    P = ring.packets_per_epoch()
    rng = np.random.default_rng(5)
    t = np.arange(n)
    got = []
    for e in range(14):
        for p in range(P):
            x = rng.normal(0, 7e-4, n) + 1j * rng.normal(0, 7e-4, n)
            if e >= 8:
                x = x + 0.02 * np.exp(2j * np.pi * 200 * (t + p * n) / n)     # band 200 // 16 = 12 driven in the later epochs
            ring.push(0, x.astype(np.complex64).view(np.float32).copy())
        ring.drain()                             # one epoch at a time: exactly six epochs feed the estimate
        got.extend(ring.poll(8))
    ring_nf, busy = ring.noise_floor()
    ring.close()
    assert len(got) == 14 and not busy
    flags = [r.flags & cs.EPOCH_CALIBRATION for r in got]
    assert flags == [1] * 6 + [0] * 8
    feats = np.array([list(r.features[:64]) for r in got], np.float32)
    nf_host = s.noise_floor_host(feats[:6])
    assert ring_nf == nf_host and all(r.noise_floor == nf_host for r in got[6:]) and all(r.noise_floor == 0 for r in got[:6])
    occ = np.array([list(r.occupancy[:64]) for r in got])
    assert not occ[:8].any()                                   # calibration epochs (+inf thresholds) and idle epochs: nothing occupied
    assert occ[8:, 12].all() and occ[8:].sum() == 6            # the driven band, and only it, against 4 x the measured floor
    s2 = cs.Sensor(cs.cfg_welch(n, K, 64))
    with pytest.raises(cs.CrnError, match="reserve"):
        s2.calibrate_thresholds(feats[:6], 4.0)                # allocates nothing: the buffers must have been reserved
    s2.reserve_noise_floor()
    assert s2.calibrate_thresholds(feats[:6], 4.0) == nf_host and s2.cfg.thresh[0] == np.float32(4.0) * np.float32(nf_host)
    s2.close()
    s.close()
