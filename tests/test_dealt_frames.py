"""The dealt-frame form of the sensing kernel (csrc/crn_sense_kernel.h: sense_kernel_dealt) — what a launch of a few epochs runs at
512 / 1024 points (no window, or the periodic Hann of the Welch plans), the engine's one-epoch launch first of all (reference: one epoch of ten 512-point frames per
sensing period, CE_Predictive_Node.cpp:148-156).  An epoch's frames are spread over the lane groups of one workgroup and the K-frame
accumulate is replayed in frame order afterwards, so every output must be BIT FOR BIT what the streaming form gives on the same
input: that equality is the test (the streaming form's parity with the oracle is everything else in tests/), plus the oracle directly
on the engine's own shape."""
import numpy as np
import pytest

import crnsense as cs
import oracle_py as orc
import signals

AUTO, NEVER, ALWAYS = 400, 401, 402          # crn_sense_set_variant codes of the dealt form


def _both(cfg, iq, n_epochs, L, want_spectrum=False, epoch_stride=0, has_dealt_form=True):
    res = []
    for code in (NEVER, ALWAYS):
        s = cs.Sensor(cfg)
        s.set_variant(code)
        res.append(s.run_host(iq, n_epochs, L=L, want_spectrum=want_spectrum, epoch_stride=epoch_stride))
        n_dealt = s.dealt_launches()
        s.close()
        assert (n_dealt > 0) == (code == ALWAYS and has_dealt_form), (code, n_dealt)
    return res


def _same(a, b, keys):
    for k in keys:
        assert np.array_equal(a[k], b[k], equal_nan=True), k


def _plans(n):
    yield "reference plan, energy", cs.cfg_energy_scaled(n, 4.0)
    yield "reference plan, |X| + network", cs.cfg_reference_scaled(n)
    c = cs.cfg_energy_scaled(n, 4.0)                      # a band outside the reference plan's rows, another that wraps around DC
    c.segs[2].lo, c.segs[2].hi = n // 4 + 3, n // 4 + 41
    c.segs[0].lo, c.segs[0].hi = 0, 5
    yield "other plan", c
    c = cs.cfg_energy_scaled(n, 4.0)                      # 16 bands of n / 16 bins: no register close, the LDS walk
    c.n_bands, c.n_segs, c.ref_band = 16, 16, -1
    for b in range(16):
        c.segs[b].lo, c.segs[b].hi, c.segs[b].band = b * (n // 16), (b + 1) * (n // 16), b
        c.thresh[b] = 1e-3
    yield "16 bands", c


@pytest.mark.gpu
@pytest.mark.parametrize("n", [512, 1024])
def test_dealt_frames_equal_the_streaming_kernel(built, n):
    for name, base in _plans(n):
        for K in (2, 3, 7, 8, 9, 10):
            for L in (n, 364, 100, 1):
                for n_epochs in (1, 5):
                    cfg = base
                    cfg.frames_per_epoch = K
                    iq, _ = signals.make_epochs(cfg, n_epochs, seed=7 * n + 13 * K + L, L=L)
                    a, b = _both(cfg, iq, n_epochs, L)
                    keys = ["features", "occupancy"] + (["ann_out", "decision"] if cfg.decide == cs.DECIDE_ANN else [])
                    _same(a, b, keys)
        # the per-bin spectrum leaves through the same close
        cfg = base
        cfg.frames_per_epoch = 10
        iq, _ = signals.make_epochs(cfg, 3, seed=n + 99, L=n)
        a, b = _both(cfg, iq, 3, n, want_spectrum=True)
        _same(a, b, ["features", "occupancy", "spectrum"])


def _windowed(n):
    yield "Welch: periodic Hann, hop N/2, 64 bands (window folded into pass 1)", cs.cfg_welch(n, 8, 64), (n,)
    yield "Welch, 16 bands", cs.cfg_welch(n, 5, 16), (n,)
    c = cs.cfg_energy_scaled(n, 4.0)
    c.window = cs.WINDOW_HANN                              # disjoint Hann frames: still the folded form on whole frames, the table on short ones
    yield "Hann, disjoint frames", c, (n, 364, 100)
    c = cs.cfg_energy_scaled(n, 4.0)
    c.window = cs.WINDOW_BLACKMAN_HARRIS
    yield "Blackman-Harris (table window)", c, (n, 364)
    c = cs.cfg_welch(n, 8, 64)
    c.mode = cs.MODE_REF_MAG
    yield "Welch on magnitudes", c, (n,)
    c = cs.cfg_reference_scaled(n)
    c.window = cs.WINDOW_HANN
    yield "reference mode behind a Hann window", c, (n, 364)


@pytest.mark.gpu
@pytest.mark.parametrize("n", [512, 1024])
def test_windowed_dealt_frames_equal_the_streaming_kernel(built, n):
    """Windows and overlapped frames (the engine's -m welch / -m scan at its default size): each lane group fetches its whole frame,
    the window is applied the way the streaming dispatch applies it for the same launch, so the outputs are again bit-identical.
    The one windowed dealt form is the periodic Hann on whole frames in energy mode (what those engine modes launch); every other
    window, |X| mode and short packets have none: forcing the dealt form changes nothing, the streaming kernel takes the launch."""
    for name, base, Ls in _windowed(n):
        dealt_form = lambda L: base.window == cs.WINDOW_HANN and base.mode == cs.MODE_ENERGY and L == n   # noqa: E731
        for K in (2, 5, 8, 10):
            for L in Ls:
                for n_epochs in (1, 4):
                    cfg = base
                    cfg.frames_per_epoch = K
                    for b in range(cfg.n_bands):
                        if cfg.decide == cs.DECIDE_THRESHOLD and cfg.ref_band < 0:
                            cfg.thresh[b] = 1e-3
                    iq, _ = signals.make_epochs(cfg, n_epochs, seed=3 * n + 11 * K + L, L=L)
                    a, b = _both(cfg, iq, n_epochs, L, has_dealt_form=dealt_form(L))
                    keys = ["features", "occupancy"] + (["ann_out", "decision"] if cfg.decide == cs.DECIDE_ANN else [])
                    _same(a, b, keys)
        cfg = base
        cfg.frames_per_epoch = 8
        iq, _ = signals.make_epochs(cfg, 3, seed=n + 7, L=n)
        a, b = _both(cfg, iq, 3, n, want_spectrum=True, has_dealt_form=dealt_form(n))
        _same(a, b, ["features", "occupancy", "spectrum"])
        want = orc.run(cfg, iq, 3, L=n)
        assert np.allclose(b["features"], want["features"], rtol=1e-5, atol=0), name


@pytest.mark.gpu
def test_dealt_frames_with_gaps_between_epochs_and_a_ragged_batch(built):
    """Epochs further apart than they are long (epoch_stride), and a batch that ends with the last epoch's last sample: the dealt
    workgroup's window is its own epoch, nothing of the next one."""
    cfg = cs.cfg_reference()
    L, K, n_epochs = 364, 10, 6
    stride = K * L + 777
    rng = np.random.default_rng(5)
    iq = rng.normal(0, 1e-2, ((n_epochs - 1) * stride + K * L) * 2).astype(np.float32)
    a, b = _both(cfg, iq, n_epochs, L, epoch_stride=stride)
    _same(a, b, ["features", "ann_out", "decision", "occupancy"])
    dense = np.concatenate([iq[2 * e * stride: 2 * (e * stride + K * L)] for e in range(n_epochs)])
    want = orc.run(cfg, dense, n_epochs, L=L)
    assert np.allclose(b["features"], want["features"], rtol=1e-5, atol=0)
    # poison behind every epoch: no frame reads past its epoch's K L samples
    iq2 = iq.copy().reshape(-1, 2)
    for e in range(n_epochs - 1):
        iq2[e * stride + K * L: (e + 1) * stride] = np.nan
    a2, b2 = _both(cfg, iq2.ravel(), n_epochs, L, epoch_stride=stride)
    _same(a, a2, ["features", "decision"])
    _same(b, b2, ["features", "decision"])


@pytest.mark.gpu
def test_the_engines_one_epoch_launch_is_dealt_and_matches_the_oracle(built):
    """Automatic choice: one reference epoch of ten 364-sample packets (the engine's launch) runs the dealt form, a batch of 4096
    epochs the streaming form; decisions, network outputs and features against the oracle either way."""
    import parity_policy as pol
    cfg = cs.cfg_reference()
    L = 364
    iq, picks = signals.make_epochs(cfg, 4096, seed=2024, L=L)
    want = orc.run(cfg, iq, 4096, L=L)
    s = cs.Sensor(cfg)
    spe = cs.samples_per_epoch(cfg, L)
    for e in range(64):
        got = s.run_host(iq[2 * e * spe: 2 * (e + 1) * spe], 1, L=L)
        assert np.allclose(got["features"][0], want["features"][e], rtol=1e-5, atol=0)
        assert np.abs(got["ann_out"][0] - want["ann_out"][e]).max() < 1e-4
        if (np.abs(want["ann_out"][e] - cfg.ann_threshold) > pol.ANN_MARGIN).all():
            assert got["decision"][0] == want["decision"][e] == picks[e]
    assert s.dealt_launches() == 64
    big = s.run_host(iq, 4096, L=L)
    assert s.dealt_launches() == 64                       # 4096 epochs: the streaming form
    for e in range(64):                                   # ... and the two forms agree bit for bit on the same epochs
        got = s.run_host(iq[2 * e * spe: 2 * (e + 1) * spe], 1, L=L)
        assert np.array_equal(got["features"][0], big["features"][e]) and got["decision"][0] == big["decision"][e]
        assert np.array_equal(got["ann_out"][0], big["ann_out"][e])
    s.close()


@pytest.mark.gpu
def test_dealt_frames_in_the_wire_format(built):
    """int16 pairs through the dealt form: bit-identical to its float path on the converted samples (as for the streaming form,
    tests/test_sc16.py), one epoch and a handful."""
    import torch
    dev = torch.device("cuda", 0)
    for cfg, L in ((cs.cfg_reference(), 364), (cs.cfg_energy_scaled(1024, 4.0), 1024)):
        for n in (1, 9):
            need = cs.samples_needed(cfg, n, L)
            rng = np.random.default_rng(n + L)
            raw = rng.integers(-3000, 3000, (need, 2), dtype=np.int16)
            host_f = (raw.astype(np.float32) / np.float32(32768.0)).ravel()
            d_raw, d_f = torch.from_numpy(raw.copy()).to(dev), torch.from_numpy(host_f).to(dev)
            res = {}
            for code in (NEVER, ALWAYS):
                s = cs.Sensor(cfg)
                s.set_variant(code)
                for sc in (False, True):
                    feats = torch.zeros(n, cfg.n_bands, device=dev)
                    ann = torch.zeros(n, 3, dtype=torch.float64, device=dev)
                    dec = torch.full((n,), -7, dtype=torch.int32, device=dev)
                    occ = torch.full((n, cfg.n_bands), 9, dtype=torch.uint8, device=dev)
                    s.run_device((d_raw if sc else d_f).data_ptr(), n, L,
                                 {"features": feats.data_ptr(), "ann_out": ann.data_ptr(), "decision": dec.data_ptr(),
                                  "occupancy": occ.data_ptr(), "spectrum": 0}, sc16=sc)
                    torch.cuda.synchronize()
                    res[code, sc] = (feats, ann, dec, occ)
                assert s.dealt_launches() == (2 if code == ALWAYS else 0)
                s.close()
            for other in ((NEVER, True), (ALWAYS, False), (ALWAYS, True)):
                for x, y in zip(res[NEVER, False], res[other]):
                    assert torch.equal(x, y), other


if not cs.has_sc16():
    # (the optional wire-format kernels are not in this library: tests/test_sc16.py runs this test against libcrnsense_sc16.so in a child process)
    del test_dealt_frames_in_the_wire_format
