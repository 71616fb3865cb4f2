"""CPU tests of the oracle (oracle/crn_oracle.c): the restated liquid-style fp32 FFT against
float64 ground truth, hand-derived known answers of the reference pipeline, and self-consistency
between the literal reference epoch and the generalised path.  No GPU."""
import numpy as np
import pytest

import crnsense as cs
import oracle_py as orc
import signals


@pytest.mark.parametrize("n", [8, 64, 512, 1024, 4096])
def test_fft_radix2_matches_float64(built, n):
    rng = np.random.default_rng(n)
    x = (rng.normal(size=n) + 1j * rng.normal(size=n)).astype(np.complex64)
    y = orc.fft_radix2(x)
    ref = np.fft.fft(x.astype(np.complex128))
    # fp32 radix-2: error grows ~ log2(n) * eps * |X|_rms
    assert np.abs(y - ref).max() <= 4e-7 * np.log2(n) * np.sqrt(n) * 3


@pytest.mark.parametrize("n", [8, 64, 512])
def test_long_double_dft_matches_numpy(built, n):
    rng = np.random.default_rng(n + 1)
    x = rng.normal(size=n) + 1j * rng.normal(size=n)
    assert np.abs(orc.dft_f64(x) - np.fft.fft(x)).max() < 1e-11 * n


@pytest.mark.parametrize("n", [512, 1024, 2048, 4096])
def test_impulse_and_dc(built, n):
    """SURVEY.md §8(c) known answers at every size: a unit impulse has |X[k]| = 1 everywhere, DC lands in bin 0 with value N."""
    x = np.zeros(n, np.complex64)
    x[0] = 1
    assert np.allclose(orc.fft_radix2(x), np.ones(n), atol=0)  # exact: only additions of 0 and 1*t
    x[:] = 1
    y = orc.fft_radix2(x)
    assert y[0] == n and np.abs(y[1:]).max() < 1e-4 * (n / 512)


def test_forward_sign(built):
    # X[k] = sum x[n] exp(-j 2 pi k n / N): a tone exp(+j 2 pi 70 n / N) lands in bin 70
    n = 512
    t = np.exp(2j * np.pi * 70 * np.arange(n) / n).astype(np.complex64)
    y = np.abs(orc.fft_radix2(t))
    assert y.argmax() == 70 and abs(y[70] - n) < 1e-2


def test_reference_weights_match_product_cfg(built):
    wih, who = orc.ref_weights()
    c = cs.cfg_reference()
    assert np.array_equal(np.ctypeslib.as_array(c.ann_w_ih), wih)
    assert np.array_equal(np.ctypeslib.as_array(c.ann_w_ho), who)
    assert c.fft_len == 512 and c.frames_per_epoch == 10 and c.ann_threshold == 0.8
    segs = [(c.segs[i].lo, c.segs[i].hi, c.segs[i].band) for i in range(c.n_segs)]
    # CE_Predictive_Node.cpp:173-191 (bin 511 excluded from CH1)
    assert segs == [(0, 16, 1), (496, 511, 1), (55, 85, 2), (189, 222, 3), (300, 310, 0)]
    assert list(c.tx_freq_for_decision) == [0.0, 835e6, 833e6, 835e6]


def test_all_zero_epoch_known_answer(built):
    # zero features -> O ~ [0.4790, 4.12e-5, 3.35e-3] -> "ALL BUSY", no tx call
    r = orc.ref_epoch(np.zeros(10 * 512 * 2, np.float32), 512)
    assert r["decision"] == 0 and r["tx_freq"] == 0.0
    assert np.all(r["features"] == 0)
    assert np.allclose(r["ann_out"], [0.4790, 4.12e-5, 3.35e-3], rtol=2e-3)
    # hand evaluation of the net in numpy double
    wih, who = orc.ref_weights()
    hid = 1 / (1 + np.exp(-wih[0, 1:]))
    out = 1 / (1 + np.exp(-(who[0, 1:] + hid @ who[1:, 1:])))
    assert np.allclose(r["ann_out"], out, rtol=1e-14)


@pytest.mark.parametrize("bin_,band", [(0, 1), (15, 1), (16, None), (495, None), (496, 1), (510, 1), (511, None),
                                       (54, None), (55, 2), (84, 2), (85, None), (188, None), (189, 3),
                                       (221, 3), (222, None), (299, None), (300, 0), (309, 0), (310, None)])
def test_band_edges_single_tone(built, bin_, band):
    """A unit on-bin tone puts |X| = 512 in exactly one bin: each frame adds 51.2 to fft_avg[bin],
    so the band containing it gets M = 512 and feature 512^2; every other feature is ~0."""
    n = np.arange(512)
    frame = np.exp(2j * np.pi * bin_ * n / 512).astype(np.complex64)
    iq = np.tile(frame, 10).view(np.float32)
    r = orc.ref_epoch(iq, 512)
    order = [0, 1, 2, 3]  # features: NF, CH1, CH2, CH3
    for b in order:
        if b == band:
            assert abs(r["features"][b] - 512.0 ** 2) / 512.0 ** 2 < 1e-5
        else:
            assert r["features"][b] < 1e-3
    assert abs(r["fft_avg"][bin_] - 512.0) < 1e-2


def scaled_edge_bins(cfg):
    """[(bin, band or None)]: for every run [lo, hi) of the band plan the bins lo - 1, lo, hi - 1, hi (mod N), each with the band that
    holds it under the plan — derived from the plan's runs themselves, so that a neighbour which falls into another run is labelled so."""
    n = cfg.fft_len
    owner = {}
    for s in range(cfg.n_segs):
        g = cfg.segs[s]
        for k in range(g.lo, g.hi):
            owner[k] = g.band
    bins = []
    for s in range(cfg.n_segs):
        g = cfg.segs[s]
        for k in (g.lo - 1, g.lo, g.hi - 1, g.hi):
            bins.append(k % n)
    return [(k, owner.get(k)) for k in dict.fromkeys(bins)]


@pytest.mark.parametrize("n", [1024, 4096])
def test_band_edges_of_the_scaled_plans(built, n):
    """The same single-tone sweep at BASELINE's own sizes (cfg1: 1024 points, headline: 4096): the reference's runs scaled by N / 512
    keep the gap below bin 0 — CH1's upper run ends 1 x N/512 bins short of N (the reference leaves out bin 511,
    CE_Predictive_Node.cpp:177).  Energy mode: a unit on-bin tone puts N^2 into exactly its band's feature."""
    cfg = cs.cfg_energy_scaled(n, 4.0)
    edges = scaled_edge_bins(cfg)
    s = n // 512
    assert (n - s, None) in edges and (n - s - 1, 1) in edges and (16 * s, None) in edges and (0, 1) in edges   # the scaled 511 gap, CH1's ends
    t = np.arange(n)
    x = np.stack([np.tile(np.exp(2j * np.pi * k * t / n), cfg.frames_per_epoch) for k, _ in edges]).astype(np.complex64)
    r = orc.run(cfg, x.view(np.float32).ravel(), len(edges))
    for i, (k, band) in enumerate(edges):
        for b in range(4):
            if b == band:
                assert abs(r["features"][i, b] / float(n) ** 2 - 1) < 1e-5, (k, b)
            else:
                assert r["features"][i, b] < 1e-6 * float(n) ** 2, (k, b, r["features"][i, b])


def test_cascade_and_tx_mapping(built):
    # CE_Predictive_Node.cpp:245-258: CH1 busy -> 835e6, CH2 busy -> 833e6, CH3 busy -> 835e6
    cfg = cs.cfg_reference()
    for pick, tx in ((1, 835e6), (2, 833e6), (3, 835e6), (0, 0.0)):
        iq, _ = signals.make_epochs(cfg, 1, seed=100 + pick, picks=[pick])
        r = orc.ref_epoch(iq, 512)
        assert r["decision"] == pick and r["tx_freq"] == tx


@pytest.mark.parametrize("L", [512, 364, 363, 1])
def test_generic_path_equals_literal_reference_epoch(built, L):
    cfg = cs.cfg_reference()
    n_epochs = 6
    iq, picks = signals.make_epochs(cfg, n_epochs, seed=7 + L, L=L)
    g = orc.run(cfg, iq, n_epochs, L=L, want_spectrum=True)
    for e in range(n_epochs):
        r = orc.ref_epoch(iq[e * 10 * L * 2:(e + 1) * 10 * L * 2], L)
        assert np.array_equal(r["fft_avg"], g["spectrum"][e])
        assert np.array_equal(r["features"], g["features"][e])
        assert np.array_equal(r["ann_out"], g["ann_out"][e])
        assert r["decision"] == g["decision"][e]


def test_oracle_spectrum_within_tolerance_of_float64(built):
    for cfg in (cs.cfg_reference(), cs.cfg_energy_scaled(1024), cs.cfg_energy_scaled(4096)):
        iq, _ = signals.make_epochs(cfg, 4, seed=11)
        g = orc.run(cfg, iq, 4, want_spectrum=True)
        truth = signals.spectrum_f64(cfg, iq, 4)
        floor = 1e-2 * truth.mean(axis=1, keepdims=True)
        err = np.abs(g["spectrum"] - truth) / np.maximum(truth, floor)
        assert err.max() < 1e-5


def test_threshold_decision_semantics(built):
    cfg = cs.cfg_energy_scaled(1024, 4.0)
    iq, picks = signals.make_epochs(cfg, 12, seed=5)
    g = orc.run(cfg, iq, 12)
    for e in range(12):
        want = np.zeros(4, np.uint8)
        if picks[e] > 0:
            want[picks[e]] = 1
        assert np.array_equal(g["occupancy"][e], want)
        assert g["decision"][e] == int(picks[e] > 0)


def test_oracle_rejects_bad_arguments(built):
    import ctypes as C
    cfg = cs.cfg_reference()
    o = cs.Out()
    iq = np.zeros(16, np.float32)
    assert orc.lib().crn_oracle_run(C.byref(cfg), iq.ctypes.data, 1, 513, 0, C.byref(o), 1) == -1
    assert orc.lib().crn_oracle_run(C.byref(cfg), iq.ctypes.data, 1, 0, 0, C.byref(o), 1) == -1


@pytest.mark.parametrize("n", [1024, 512])
def test_cfg0_thousand_epochs_cpu_plumbing(built, n):
    """BASELINE.json configs[0] / SURVEY.md §8d cfg0: 3 channels + noise floor, K = 10, one stream, 1000
    epochs, CPU restatement only: every decision follows the driven occupancy, with the generated
    traffic of the generator twin (uniform model) as input.  N = 1024 decides by threshold against the
    noise-floor band, N = 512 is the reference configuration with its network."""
    import oracle_py as orc
    cfg = cs.cfg_energy_scaled(1024, 4.0) if n == 1024 else cs.cfg_reference()
    sc = cs.SynthCfg()
    sc.seed, sc.noise_power, sc.signal_rms, sc.tones_per_band = 0xC0FFEE, 1e-6, 0.02, 8
    sc.pu_model, sc.signal_kind, sc.n_streams = cs.PU_UNIFORM, cs.SIG_TONES, 1
    n_epochs = 1000
    iq, truth = orc.synth(cfg, sc, n_epochs, cs.samples_per_epoch(cfg))
    got = orc.run(cfg, iq, n_epochs, n_threads=4)
    if n == 512:
        assert np.array_equal(got["decision"], truth)
    else:
        want = np.zeros((n_epochs, 4), np.uint8)
        idx = np.nonzero(truth > 0)[0]
        want[idx, truth[idx]] = 1
        assert np.array_equal(got["occupancy"], want)
    assert set(truth.tolist()) == {0, 1, 2, 3}
