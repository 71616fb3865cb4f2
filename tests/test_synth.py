"""Signal generator: the CPU twin's traffic models and signal kinds (no GPU), and
the device generator against the twin sample by sample (gpu)."""
import numpy as np
import pytest

import crnsense as cs
import oracle_py as orc


def _sc(seed=7, pu=cs.PU_UNIFORM, sig=cs.SIG_TONES, tones=8, n_streams=1, noise=1e-6, rms=0.02):
    sc = cs.SynthCfg()
    sc.seed, sc.noise_power, sc.signal_rms = seed, noise, rms
    sc.tones_per_band, sc.pu_model, sc.signal_kind, sc.n_streams = tones, pu, sig, n_streams
    return sc


def test_markov_as_written_never_reaches_ch3(built):
    """CE_PU_MARKOV_Chain_Tx.cpp:96-123: `>= 1 || < 4` is always true, so outcome 0 -> CH1 and every
    other outcome -> CH2, whatever the state."""
    cfg = cs.cfg_reference()
    n_streams, eps = 8, 4000
    _, truth = orc.synth(cfg, _sc(pu=cs.PU_MARKOV_AS_WRITTEN, n_streams=n_streams, noise=0.0, tones=1), n_streams * eps, 1)
    assert set(truth.tolist()) == {1, 2}
    assert abs((truth == 1).mean() - 0.1) < 0.01


def test_markov_intended_stationary_distribution(built):
    """P = [[.1,.3,.6],[.1,.5,.4],[.1,.3,.6]] -> stationary (0.1, 0.375, 0.525)."""
    cfg = cs.cfg_reference()
    n_streams, eps = 16, 8000
    _, truth = orc.synth(cfg, _sc(pu=cs.PU_MARKOV_INTENDED, n_streams=n_streams, noise=0.0, tones=1), n_streams * eps, 1)
    t = truth.reshape(n_streams, eps)
    assert (t[:, 0] >= 1).all() and set(truth.tolist()) == {1, 2, 3}
    freq = np.array([(truth == k).mean() for k in (1, 2, 3)])
    assert np.abs(freq - np.array([0.1, 0.375, 0.525])).max() < 0.01
    # transition rows
    P = np.zeros((3, 3))
    for a, b in zip(t[:, :-1].ravel(), t[:, 1:].ravel()):
        P[a - 1, b - 1] += 1
    P /= P.sum(axis=1, keepdims=True)
    assert np.abs(P - np.array([[.1, .3, .6], [.1, .5, .4], [.1, .3, .6]])).max() < 0.02
    # streams are independent chains
    assert len({tuple(r[:64]) for r in t}) == n_streams


def test_uniform_model_includes_idle(built):
    cfg = cs.cfg_reference()
    _, truth = orc.synth(cfg, _sc(noise=0.0, tones=1), 20000, 1)
    freq = np.bincount(truth, minlength=4) / truth.size
    assert np.abs(freq - 0.25).max() < 0.02


@pytest.mark.parametrize("sig", [cs.SIG_TONES, cs.SIG_CW, cs.SIG_BAND_NOISE])
def test_signal_kinds_land_in_their_band(built, sig):
    """Spectrum of the noiseless signal: CW = one bin at the band centre, tones = tones_per_band bins,
    band noise = every bin of the band at equal power; total power = signal_rms^2 in all three."""
    cfg = cs.cfg_energy_scaled(1024, 4.0)
    spe = cs.samples_per_epoch(cfg)
    n_epochs = 40
    iq, truth = orc.synth(cfg, _sc(sig=sig, noise=0.0, tones=6, rms=0.5), n_epochs, spe)
    x = iq.view(np.complex64).reshape(n_epochs, cfg.frames_per_epoch, cfg.fft_len)
    for e in range(n_epochs):
        P = (np.abs(np.fft.fft(x[e], axis=1)) ** 2).mean(axis=0) / cfg.fft_len ** 2
        if truth[e] == 0:
            assert P.sum() == 0
            continue
        bins = np.concatenate([np.arange(cfg.segs[s].lo, cfg.segs[s].hi) for s in range(cfg.n_segs)
                               if cfg.segs[s].band == truth[e]])
        inband = P[bins].sum()
        assert abs(inband - 0.25) < 1e-3 and abs(P.sum() - inband) < 1e-6
        lit = (P > 1e-9).sum()
        assert lit == {cs.SIG_TONES: 6, cs.SIG_CW: 1, cs.SIG_BAND_NOISE: bins.size}[sig]
        if sig == cs.SIG_BAND_NOISE:
            assert np.allclose(P[bins], 0.25 / bins.size, rtol=1e-3)
    assert len(set(truth.tolist())) == 4


def _band_bins(cfg, band):
    return np.concatenate([np.arange(cfg.segs[s].lo, cfg.segs[s].hi) for s in range(cfg.n_segs) if cfg.segs[s].band == band])


@pytest.mark.parametrize("sig", [cs.SIG_RRC_QPSK, cs.SIG_GMSK, cs.SIG_OFDM])
@pytest.mark.parametrize("plan", ["energy1024", "ref512"])
def test_modulated_carriers_fill_their_band(built, sig, plan):
    """The three continuous waveforms of src/interferer.cpp:160-282 (not aligned to the FFT grid): total power signal_rms^2, nearly
    all of it inside the driven band (a long Hann periodogram over the whole epoch, finer than the sensing grid), the band
    actually filled (no half-empty band), GMSK of constant envelope, and the reference plan's CH1 — which wraps around DC
    (bins 496..510 and 0..15) — centred on its signed centre."""
    cfg = cs.cfg_energy_scaled(1024, 4.0) if plan == "energy1024" else cs.cfg_reference()
    N = cfg.fft_len
    spe = cs.samples_per_epoch(cfg)
    n_epochs = 24
    rms = 0.5
    iq, truth = orc.synth(cfg, _sc(sig=sig, noise=0.0, rms=rms, pu=cs.PU_SWEEP, n_streams=2), n_epochs, spe)
    x = iq.view(np.complex64).reshape(n_epochs, spe).astype(np.complex128)
    assert truth.tolist() == [1, 2, 3, 2, 1, 2, 3, 2, 1, 2, 3, 2] * 2          # interferer.cpp:339-345, restarted per stream
    power = (np.abs(x) ** 2).mean(axis=1)
    assert np.abs(power / rms ** 2 - 1).max() < (1e-6 if sig == cs.SIG_GMSK else 0.12)     # one epoch is a few hundred symbols
    assert abs(power.mean() / rms ** 2 - 1) < 0.03
    if sig == cs.SIG_GMSK:
        assert np.abs(np.abs(x) - rms).max() < 1e-6
    w = np.hanning(spe)
    f = np.fft.fftfreq(spe) * N                                                  # in sensing bins
    skew = []
    for e in range(n_epochs):
        bins = _band_bins(cfg, int(truth[e])).astype(np.int64)
        signed = np.where(bins >= N // 2, bins - N, bins)
        lo, hi = signed.min() - 0.5, signed.max() + 0.5
        P = np.abs(np.fft.fft(x[e] * w)) ** 2
        inband = P[(f >= lo) & (f <= hi)].sum() / P.sum()
        assert inband > {cs.SIG_RRC_QPSK: 0.995, cs.SIG_GMSK: 0.995, cs.SIG_OFDM: 0.93}[sig], (e, inband)
        # the inner 60 % of the band holds its share: the band is filled, not a narrow line in it
        c, wdt = 0.5 * (lo + hi), hi - lo
        inner = P[np.abs(f - c) <= 0.3 * wdt].sum() / P.sum()
        assert 0.5 < inner < (0.99 if sig == cs.SIG_GMSK else 0.9), (e, inner)   # GMSK at BT 0.5 is compact: 99 % inside 1.04 Rs
        # symmetric about the band centre (a carrier off by a bin or with the wrong sign fails this)
        left = P[(f >= lo) & (f < c)].sum() / P.sum()
        assert abs(left - 0.5 * inband) < 0.25, (e, left, inband)              # one epoch is a few hundred random symbols
        skew.append(left - 0.5 * inband)
    assert abs(np.mean(skew)) < 0.04, np.mean(skew)


@pytest.mark.parametrize("sig", [cs.SIG_RRC_QPSK, cs.SIG_GMSK, cs.SIG_OFDM])
def test_oracle_occupancy_follows_modulated_traffic(built, sig):
    """Energy detection (N = 1024, threshold 4 x the noise band) of the modulated carriers over noise: exactly the driven band."""
    cfg = cs.cfg_energy_scaled(1024, 4.0)
    spe = cs.samples_per_epoch(cfg)
    n = 48
    iq, truth = orc.synth(cfg, _sc(sig=sig, pu=cs.PU_MARKOV_INTENDED, n_streams=4), n, spe)
    got = orc.run(cfg, iq, n)
    want = np.zeros((n, cfg.n_bands), np.uint8)
    want[np.arange(n), truth] = 1
    assert np.array_equal(got["occupancy"], want)


def test_rrc_pulse_is_a_nyquist_root(built):
    """What makes the QPSK carrier "RRC": the symbol pulse convolved with itself is zero at every other symbol instant.  Recovered from
    the generator itself: one band, noise off, a matched filter built from the same formula in numpy, symbols read back at the
    symbol instants with no inter-symbol interference."""
    cfg = cs.cfg_energy_scaled(1024, 4.0)
    N, spe = cfg.fft_len, cs.samples_per_epoch(cfg)
    iq, truth = orc.synth(cfg, _sc(sig=cs.SIG_RRC_QPSK, noise=0.0, rms=1.0, pu=cs.PU_SWEEP), 1, spe)
    x = iq.view(np.complex64).astype(np.complex128)
    bins = _band_bins(cfg, int(truth[0]))
    signed = np.where(bins >= N // 2, bins - N, bins)                          # CH1 wraps around DC
    centre = 0.5 * (signed.min() + signed.max())
    x = x * np.exp(-2j * np.pi * centre * np.arange(spe) / N)               # back to baseband
    beta, sps = 0.35, N * 1.35 / bins.size

    def rrc(t):
        t = np.asarray(t, np.float64)
        q = 4 * beta * t
        with np.errstate(divide="ignore", invalid="ignore"):
            h = (np.sin(np.pi * t * (1 - beta)) + q * np.cos(np.pi * t * (1 + beta))) / (np.pi * t * (1 - q * q))
        h[np.abs(t) < 1e-9] = 1 - beta + 4 * beta / np.pi
        return h

    n = np.arange(spe)
    got = []
    for k in range(20, 60):                                                    # matched filter output at symbol instant k
        h = rrc((n - k * sps) / sps)
        h[np.abs(n - k * sps) > 8 * sps] = 0
        got.append((x * h).sum() / sps)
    got = np.array(got) * np.sqrt(2)
    assert np.abs(np.abs(got.real) - 1).max() < 0.02 and np.abs(np.abs(got.imag) - 1).max() < 0.02   # (+-1 +-j): no ISI
    assert len({(int(np.sign(g.real)), int(np.sign(g.imag))) for g in got}) == 4


@pytest.mark.parametrize("pu", [cs.PU_UNIFORM, cs.PU_MARKOV_INTENDED])
def test_oracle_decisions_follow_generated_traffic(built, pu):
    """Reference-mode sensing of generated traffic: the cascade reports exactly the driven channel."""
    cfg = cs.cfg_reference()
    spe = cs.samples_per_epoch(cfg)
    n = 96
    iq, truth = orc.synth(cfg, _sc(pu=pu, n_streams=4), n, spe)
    got = orc.run(cfg, iq, n)
    assert np.array_equal(got["decision"], truth)


def test_adc_quantisation_of_the_twin(built):
    """adc_bits = 16: the samples UHD hands the reference's engine (16-bit integers from the wire scaled to floats): every component
    is a multiple of 2^-15 within half a step of the unquantised sample, clipped to [-1, 1); sensing decisions do not change."""
    cfg = cs.cfg_reference()
    spe = cs.samples_per_epoch(cfg)
    n = 32
    sc = _sc(seed=21, pu=cs.PU_UNIFORM)
    full, truth = orc.synth(cfg, sc, n, spe)
    sc.adc_bits = 16
    q, truth_q = orc.synth(cfg, sc, n, spe)
    assert np.array_equal(truth, truth_q)
    k = q.astype(np.float64) * 32768.0
    assert np.array_equal(k, np.round(k)) and np.abs(k).max() < 32768
    assert np.abs(q.astype(np.float64) - full).max() <= 0.5 / 32768 + 1e-9
    assert len(np.unique(q)) < 4000 < len(np.unique(full))
    assert np.array_equal(orc.run(cfg, q, n)["decision"], truth)
    sc.adc_bits, sc.signal_rms = 4, 3.0                                   # a coarse, overdriven converter clips
    c, _ = orc.synth(cfg, sc, n, spe)
    assert c.max() == 0.875 and c.min() == -1.0
    import ctypes as C
    sc.adc_bits = 1
    assert orc.lib().crn_oracle_synth(C.byref(cfg), C.byref(sc), q.ctypes.data, n, spe, truth.ctypes.data) == -1


def test_oracle_synth_rejects_bad_arguments(built):
    import ctypes as C
    cfg = cs.cfg_reference()
    sc = _sc(pu=cs.PU_MARKOV_INTENDED, n_streams=5)
    iq = np.zeros(12 * 2, np.float32)
    truth = np.zeros(12, np.int32)
    assert orc.lib().crn_oracle_synth(C.byref(cfg), C.byref(sc), iq.ctypes.data, 12, 1, truth.ctypes.data) == -1


# ---- device generator against the twin ------------------------------------------------------

@pytest.mark.gpu
@pytest.mark.parametrize("sig", [cs.SIG_RRC_QPSK, cs.SIG_GMSK, cs.SIG_OFDM])
@pytest.mark.parametrize("mode", ["ref512", "energy1024", "welch4096"])
def test_device_modulated_carriers_match_twin(built, sig, mode):
    """The modulated carriers on the device against the CPU twin sample by sample (both evaluate the pulses in double), the sweep's
    truth, and the sensing kernel's occupancy of the generated samples (energy plans)."""
    import torch
    dev = torch.device("cuda", 0)
    cfg = {"ref512": cs.cfg_reference, "energy1024": lambda: cs.cfg_energy_scaled(1024, 4.0), "welch4096": lambda: cs.cfg_welch(4096, 10, 64)}[mode]()
    spe = cs.samples_per_epoch(cfg)
    n_streams, eps = 3, 7
    n = n_streams * eps
    sc = _sc(seed=2000 + sig, pu=cs.PU_SWEEP, sig=sig, n_streams=n_streams)
    sc.adc_bits = 16 if mode == "welch4096" else 0                      # one of the three plans through the 16-bit "radio"
    s = cs.Sensor(cfg)
    iq = torch.zeros(n * spe * 2, dtype=torch.float32, device=dev)
    truth = torch.full((n,), -1, dtype=torch.int32, device=dev)
    s.synth_fill_device_ex(iq.data_ptr(), n, spe, sc, truth_ptr=truth.data_ptr())
    torch.cuda.synchronize()
    want_iq, want_truth = orc.synth(cfg, sc, n, spe)
    assert np.array_equal(truth.cpu().numpy(), want_truth)
    scale = sc.signal_rms + np.sqrt(sc.noise_power)
    d = np.abs(iq.cpu().numpy() - want_iq)
    if sc.adc_bits:   # a sample within rounding of a half step may land on either neighbour: rare, and exactly one step apart
        step = 2.0 ** -(sc.adc_bits - 1)
        off = d > 1e-5 * scale
        assert off.mean() < 1e-3 and np.allclose(d[off], step, rtol=1e-6)
    else:
        assert d.max() < 1e-5 * scale
    if mode == "energy1024":
        got = s.run_host(iq.cpu().numpy(), n)
        want_occ = np.zeros((n, cfg.n_bands), np.uint8)
        want_occ[np.arange(n), want_truth] = 1
        assert np.array_equal(got["occupancy"], want_occ)
    s.close()


@pytest.mark.gpu
@pytest.mark.parametrize("pu", [cs.PU_UNIFORM, cs.PU_MARKOV_AS_WRITTEN, cs.PU_MARKOV_INTENDED])
@pytest.mark.parametrize("sig", [cs.SIG_TONES, cs.SIG_CW, cs.SIG_BAND_NOISE])
@pytest.mark.parametrize("mode", ["ref512", "energy1024"])
def test_device_generator_matches_twin(built, pu, sig, mode):
    import torch
    dev = torch.device("cuda", 0)
    cfg = cs.cfg_reference() if mode == "ref512" else cs.cfg_energy_scaled(1024, 4.0)
    spe = cs.samples_per_epoch(cfg)
    n_streams, eps = 6, 11
    n = n_streams * eps
    sc = _sc(seed=1000 + 10 * pu + sig, pu=pu, sig=sig, n_streams=n_streams)
    s = cs.Sensor(cfg)
    iq = torch.zeros(n * spe * 2, dtype=torch.float32, device=dev)
    truth = torch.full((n,), -1, dtype=torch.int32, device=dev)
    s.synth_fill_device_ex(iq.data_ptr(), n, spe, sc, truth_ptr=truth.data_ptr())
    torch.cuda.synchronize()
    want_iq, want_truth = orc.synth(cfg, sc, n, spe)
    assert np.array_equal(truth.cpu().numpy(), want_truth)
    scale = sc.signal_rms + np.sqrt(sc.noise_power)
    assert np.abs(iq.cpu().numpy() - want_iq).max() < 1e-5 * scale
    # and the sensing path on the generated samples reports the driven traffic
    got = s.run_host(iq.cpu().numpy(), n)
    if mode == "ref512":
        # the shipped weights are calibrated for one signal shape and gain: only the
        # multi-tone signal at its default level is expected to drive the cascade correctly
        if sig == cs.SIG_TONES:
            assert np.array_equal(got["decision"], want_truth)
    else:
        want_occ = np.zeros((n, cfg.n_bands), np.uint8)
        idx = np.nonzero(want_truth > 0)[0]
        want_occ[idx, want_truth[idx]] = 1
        assert np.array_equal(got["occupancy"], want_occ)
    s.close()


@pytest.mark.gpu
def test_device_generator_argument_errors(built):
    import torch
    dev = torch.device("cuda", 0)
    cfg = cs.cfg_reference()
    s = cs.Sensor(cfg)
    iq = torch.zeros(10 * 5120 * 2, dtype=torch.float32, device=dev)
    truth = torch.zeros(10, dtype=torch.int32, device=dev)
    with pytest.raises(cs.CrnError):   # Markov model without a truth buffer
        s.synth_fill_device_ex(iq.data_ptr(), 10, 5120, _sc(pu=cs.PU_MARKOV_INTENDED))
    with pytest.raises(cs.CrnError):   # streams must divide the epochs
        s.synth_fill_device_ex(iq.data_ptr(), 10, 5120, _sc(pu=cs.PU_MARKOV_INTENDED, n_streams=3), truth_ptr=truth.data_ptr())
    with pytest.raises(cs.CrnError):
        s.synth_fill_device_ex(iq.data_ptr(), 10, 5120, _sc(sig=9), truth_ptr=truth.data_ptr())
    s.close()
