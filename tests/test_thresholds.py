"""Noise-floor estimate and threshold update (thr_b = lambda x NF_est, NF_est = the median band energy):
crn_noise_floor_device against numpy, crn_sense_set_thresholds against the oracle run with the same thresholds, and its ordering on
the launch stream."""
import ctypes as C

import numpy as np
import pytest

import crnsense as cs
import oracle_py as orc
import parity_policy as pol


def _lower_median(a, axis=None):
    a = np.sort(np.asarray(a), axis=axis)
    if axis is None:
        return a.ravel()[(a.size - 1) // 2]
    return np.take(a, (a.shape[axis] - 1) // 2, axis=axis)


def test_threshold_api_argument_errors(built):
    L = cs.lib()
    nf = C.c_float()
    assert L.crn_noise_floor_device(None, None, 1, C.byref(nf), None) == cs.CRN_ERR_ARG
    assert L.crn_sense_set_thresholds(None, None, 4, None) == cs.CRN_ERR_ARG


@pytest.mark.gpu
@pytest.mark.parametrize("n_epochs,n_bands", [(1, 64), (7, 64), (4096, 64), (5000, 64), (300, 4), (33, 1), (50, 80)])
def test_noise_floor_matches_numpy(built, n_epochs, n_bands):
    import torch
    rng = np.random.default_rng(n_epochs * 100 + n_bands)
    feat = rng.gamma(4.0, 1e-3, (n_epochs, n_bands)).astype(np.float32)
    feat[rng.random(feat.shape) < 0.2] *= 300.0                        # occupied bands
    feat[:, : n_bands // 3] = np.round(feat[:, : n_bands // 3], 3)     # ties
    cfg = cs.cfg_reference()
    cfg.decide, cfg.ref_band, cfg.n_bands = cs.DECIDE_NONE, -1, n_bands
    for s in range(cfg.n_segs):
        cfg.segs[s].band = min(cfg.segs[s].band, n_bands - 1)
    sn = cs.Sensor(cfg)
    d = torch.from_numpy(feat).cuda()
    got = sn.noise_floor(d.data_ptr(), n_epochs)
    want = _lower_median(_lower_median(feat[:4096], axis=1))
    assert got == want
    if n_epochs == 7:   # a NaN among the features: no median exists, and the estimate says so
        feat[3, 5] = np.nan
        assert np.isnan(sn.noise_floor(torch.from_numpy(feat).cuda().data_ptr(), n_epochs))
    sn.close()


@pytest.mark.gpu
def test_calibrated_thresholds_match_oracle_and_find_the_driven_band(built):
    """cfg2 as SURVEY.md §8(d) words it: sense once, NF_est = median band energy, thresholds = 4 x NF_est, sense again — occupancy
    equals the oracle's with the same thresholds and flags the driven band of every epoch; before the update (thresholds = +inf)
    nothing is flagged."""
    import torch
    cfg = cs.cfg_welch(4096, 8, 64)
    for b in range(64):
        cfg.thresh[b] = float("inf")
    spe = cs.samples_per_epoch(cfg)
    n = 40
    sc = cs.SynthCfg()
    sc.seed, sc.noise_power, sc.signal_rms, sc.tones_per_band, sc.pu_model, sc.signal_kind, sc.n_streams = 11, 1e-6, 0.02, 8, cs.PU_UNIFORM, cs.SIG_OFDM, 1
    need = cs.samples_needed(cfg, n)
    dev = torch.device("cuda", 0)
    iq = torch.zeros(need * 2, dtype=torch.float32, device=dev)
    truth = torch.zeros(n, dtype=torch.int32, device=dev)
    feats = torch.empty(n, 64, dtype=torch.float32, device=dev)
    occ = torch.empty(n, 64, dtype=torch.uint8, device=dev)
    outs = {"features": feats.data_ptr(), "ann_out": 0, "decision": 0, "occupancy": occ.data_ptr(), "spectrum": 0}
    sn = cs.Sensor(cfg)
    sn.synth_fill_device_ex(iq.data_ptr(), n, spe, sc, truth_ptr=truth.data_ptr())
    sn.run_device(iq.data_ptr(), n, 4096, outs)
    torch.cuda.synchronize()
    assert occ.sum().item() == 0
    nf = sn.noise_floor(feats.data_ptr(), n)
    f = feats.cpu().numpy()
    assert nf == _lower_median(_lower_median(f, axis=1))
    expect = 64 * 4096 * 1e-6 * 0.375                                   # a band of 64 bins of Hann-windowed noise of power 1e-6
    # at or a little above the noise's own level: the untapered OFDM symbol transitions (and the change of driven band inside the
    # frame that straddles two epochs) splatter ~ -50 dB of the signal over every band, which at +26 dB SNR is a visible
    # fraction of a band's noise — the reason to estimate the floor instead of assuming it
    assert 0.97 < nf / expect < 1.7
    sn.set_thresholds([np.float32(4.0) * np.float32(nf)] * 64)
    sn.run_device(iq.data_ptr(), n, 4096, outs)
    torch.cuda.synchronize()
    want = orc.run(sn.cfg, iq.cpu().numpy(), n)                        # sn.cfg carries the new thresholds
    got = occ.cpu().numpy()
    safe = np.abs(want["features"] / (4.0 * nf) - 1) > pol.THRESHOLD_MARGIN   # 10 x the measured disagreement band (tests/test_decision_band.py)
    assert np.array_equal(got[safe], want["occupancy"][safe])
    t = truth.cpu().numpy()
    driven = t > 0
    assert driven.sum() > 10 and (got[np.nonzero(driven)[0], t[driven] - 1] == 1).all()   # pick p drives band p - 1 in the Welch plan
    assert got[~driven].sum() == 0                                                        # idle epochs: nothing above 4 x the floor
    sn.close()


@pytest.mark.gpu
def test_threshold_update_is_ordered_on_the_launch_stream(built):
    """launch, set_thresholds, launch — all on one stream, no synchronisation in between: the first launch decides with the old
    thresholds, the second with the new ones."""
    import torch
    cfg = cs.cfg_energy_scaled(1024, 4.0)
    cfg.ref_band = -1
    for b in range(cfg.n_bands):
        cfg.thresh[b] = float("inf")
    n = 20000                                                           # long enough that the copy would overtake an unordered launch
    need = cs.samples_needed(cfg, n)
    dev = torch.device("cuda", 0)
    iq = torch.randn(need * 2, dtype=torch.float32, device=dev) * 1e-3
    occ_a = torch.full((n, cfg.n_bands), 9, dtype=torch.uint8, device=dev)
    occ_b = torch.full((n, cfg.n_bands), 9, dtype=torch.uint8, device=dev)
    feats = torch.empty(n, cfg.n_bands, dtype=torch.float32, device=dev)
    st = torch.cuda.Stream()
    sn = cs.Sensor(cfg)
    with torch.cuda.stream(st):
        outs = {"features": feats.data_ptr(), "ann_out": 0, "decision": 0, "occupancy": occ_a.data_ptr(), "spectrum": 0}
        sn.run_device(iq.data_ptr(), n, 1024, outs, stream=st.cuda_stream)
        sn.set_thresholds([0.0] * cfg.n_bands, stream=st.cuda_stream)
        outs["occupancy"] = occ_b.data_ptr()
        sn.run_device(iq.data_ptr(), n, 1024, outs, stream=st.cuda_stream)
    st.synchronize()
    assert occ_a.sum().item() == 0 and occ_b.min().item() == 1
    with pytest.raises(cs.CrnError):
        sn.set_thresholds([0.0] * (cfg.n_bands + 1))
    sn.close()
