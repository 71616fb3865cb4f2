"""The C oracle against the independent float64 restatement (tests/ref_f64.py) on RANDOM plans — beyond the committed fixtures: any
band table (several runs per band, overlapping bands, a band that wraps around DC), magnitude or energy mode, rectangular / Hann /
Blackman-Harris window, Welch hop or disjoint frames, short packets, K from 1 to 12, absolute or reference-band thresholds.
Features to 1e-5 (the parity bar), occupancy exactly wherever float64 leaves a margin.  The CPU cases draw fresh examples every run;
the GPU cases run a fixed (derandomized) set, so that the round-end GPU suite is the same suite every time — fresh random
configurations on the GPU are tests/soak_gpu.py's job (seeds stated, profiles/r04_soak.txt)."""
import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

import crnsense as cs
import oracle_py as orc
import ref_f64


# a NaN limit or margin would make a comparison pass vacuously: numpy's invalid-value warnings are errors in this module
pytestmark = pytest.mark.filterwarnings("error::RuntimeWarning")


@st.composite
def plans(draw):
    n = draw(st.sampled_from([512, 1024]))
    k = draw(st.integers(1, 12))
    mode = draw(st.sampled_from(["mag", "energy"]))
    win = draw(st.sampled_from(["rect", "rect", "hann", "bh"]))
    hop = n // 2 if (win != "rect" and draw(st.booleans())) else n
    L = n if (win != "rect" or hop != n) else draw(st.sampled_from([n, n, 364, 100, 1]))
    nb = draw(st.integers(1, 6))
    runs = {}
    for b in range(nb):
        rr = []
        for _ in range(draw(st.integers(1, 3))):
            lo = draw(st.integers(0, n - 1))
            hi = draw(st.integers(lo, min(n, lo + 90)))
            rr.append((lo, hi))
        runs[b] = tuple(rr)
    ref_band = draw(st.sampled_from([-1, -1, 0]))
    thr = [draw(st.sampled_from([0.5, 2.0, 4.0, float("inf")])) if ref_band >= 0 else draw(st.sampled_from([1e-6, 1e-4, 1e-2, float("inf")]))
           for _ in range(nb)]
    seed = draw(st.integers(0, 2 ** 31 - 1))
    return n, k, mode, win, hop, L, runs, ref_band, thr, seed


def _build(p):
    n, k, mode, win, hop, L, runs, ref_band, thr, seed = p
    plan = ref_f64.Plan(n=n, k=k, hop=0 if hop == n else hop, mode=mode, window=win, runs=runs, decide="threshold", thresh=tuple(thr),
                        ref_band=ref_band)
    cfg = cs.cfg_energy_scaled(n, 4.0)
    cfg.frames_per_epoch, cfg.hop = k, hop
    cfg.mode = cs.MODE_REF_MAG if mode == "mag" else cs.MODE_ENERGY
    cfg.window = {"rect": cs.WINDOW_RECT, "hann": cs.WINDOW_HANN, "bh": cs.WINDOW_BLACKMAN_HARRIS}[win]
    cfg.decide, cfg.ref_band, cfg.n_bands = cs.DECIDE_THRESHOLD, ref_band, len(runs)
    segs = [(lo, hi, b) for b, rr in runs.items() for lo, hi in rr]
    cfg.n_segs = len(segs)
    for i, (lo, hi, b) in enumerate(segs):
        cfg.segs[i].lo, cfg.segs[i].hi, cfg.segs[i].band = lo, hi, b
    for b in range(len(runs)):
        cfg.thresh[b] = thr[b]
    n_epochs = 3
    need = cs.samples_needed(cfg, n_epochs, L)
    rng = np.random.default_rng(seed)
    iq = rng.normal(0, 1e-2, need * 2).astype(np.float32)
    tone = rng.integers(0, n)
    t = np.arange(need)
    # a carrier 13 dB above the noise bins, not more: the per-bin bar is for noise-dominated spectra (an fp32 transform's
    # error scales with the frame's total power, so a bin 60 dB under a carrier has no meaningful relative error)
    iq[0::2] += (2e-3 * np.cos(2 * np.pi * tone * t / n)).astype(np.float32)
    iq[1::2] += (2e-3 * np.sin(2 * np.pi * tone * t / n)).astype(np.float32)
    return cfg, plan, iq, n_epochs, L, runs, ref_band, thr, k


def _check(got, want, runs, ref_band, thr, k):
    scale = np.maximum(np.abs(want["features"]), 1e-30)
    empty = np.array([sum(hi - lo for lo, hi in runs[b]) == 0 for b in range(len(runs))])
    assert (np.abs(got["features"] - want["features"]) / scale)[:, ~empty].max(initial=0.0) < 1e-5
    assert np.all(got["features"][:, empty] == 0)
    # per-bin floor as everywhere: 1e-2 of the mean bin once a few frames are averaged, 1e-1 below that
    floor = (1e-2 if k >= 4 else 1e-1) * want["spectrum"].mean(axis=1, keepdims=True)
    # (a single fp32 frame leaves up to ~1.2e-5 on a weak bin; the 1e-5 bar is for averaged epochs — the reference's K is 10)
    assert (np.abs(got["spectrum"] - want["spectrum"]) / np.maximum(want["spectrum"], floor)).max() < (1e-5 if k >= 4 else 3e-5)
    thr_arr = np.asarray(thr, np.float64)[None, :]
    # the limit as ref_f64.run forms it: an infinite threshold stays infinite (inf x 0 over an empty reference band would be NaN,
    # and every comparison with a NaN limit is vacuous); rows at lim = inf or lim = 0 are decided whatever the rounding
    with np.errstate(invalid="ignore", divide="ignore"):
        lim = np.where(np.isinf(thr_arr), np.inf, thr_arr * (want["features"][:, ref_band:ref_band + 1] if ref_band >= 0 else 1.0))
        safe = np.isinf(lim) | (lim == 0) | (np.abs(want["features"] / lim - 1.0) > 1e-4)
    assert not np.isnan(lim).any()
    assert np.array_equal(got["occupancy"][safe], want["occupancy"][safe])


@settings(max_examples=200, deadline=None)
@given(plans())
def test_oracle_matches_float64_on_random_plans(built, p):
    cfg, plan, iq, n_epochs, L, runs, ref_band, thr, k = _build(p)
    _check(orc.run(cfg, iq, n_epochs, L=L, want_spectrum=True), ref_f64.run(plan, iq, n_epochs, L=L), runs, ref_band, thr, k)


@pytest.mark.gpu
@settings(max_examples=150, deadline=None, derandomize=True, suppress_health_check=[HealthCheck.function_scoped_fixture])
@given(plans())
def test_gpu_matches_float64_on_random_plans(built, p):
    """The HIP path through the C ABI against the same independent float64 restatement, same random plans, same bars."""
    cfg, plan, iq, n_epochs, L, runs, ref_band, thr, k = _build(p)
    s = cs.Sensor(cfg)
    got = s.run_host(iq, n_epochs, L=L, want_spectrum=True)
    s.close()
    _check(got, ref_f64.run(plan, iq, n_epochs, L=L), runs, ref_band, thr, k)


# ---- reference mode (|X| mean, square of sum, 4-5-3 network, cascade) on random traffic ------------------------------------------

@st.composite
def ref_traffic(draw):
    L = draw(st.sampled_from([512, 364, 363]))
    n_epochs = draw(st.integers(1, 6))
    seed = draw(st.integers(0, 2 ** 31 - 1))
    # per epoch and channel: a carrier amplitude from "absent" to "strong" (the network's whole operating range)
    amps = [[draw(st.sampled_from([0.0, 0.0, 1e-3, 3e-3, 1e-2, 3e-2])) for _ in range(3)] for _ in range(n_epochs)]
    return L, n_epochs, seed, amps


def _ref_case(t):
    L, n_epochs, seed, amps = t
    rng = np.random.default_rng(seed)
    x = (rng.normal(0, 7.07e-4, (n_epochs, 10, L)) + 1j * rng.normal(0, 7.07e-4, (n_epochs, 10, L))).astype(np.complex64)
    centres = (4, 70, 205)                                              # inside CH1 / CH2 / CH3 (REF_RUNS_512)
    n = np.arange(L)
    for e in range(n_epochs):
        for c in range(3):
            ph = rng.uniform(0, 2 * np.pi)
            x[e] += (amps[e][c] * np.exp(1j * (2 * np.pi * centres[c] * n / 512 + ph))).astype(np.complex64)
    iq = np.ascontiguousarray(x).view(np.float32).ravel()
    return iq, n_epochs, L


def _check_ref(got, want):
    assert (np.abs(got["features"] - want["features"]) / np.abs(want["features"])).max() < 1e-5
    assert np.abs(got["ann_out"] - want["ann_out"]).max() < 1e-4        # fp32 features through a steep network
    sure = want["margin"] > 1e-3
    assert np.array_equal(got["decision"][sure], want["decision"][sure])


@settings(max_examples=150, deadline=None)
@given(ref_traffic())
def test_oracle_reference_mode_on_random_traffic(built, t):
    iq, n_epochs, L = _ref_case(t)
    _check_ref(orc.run(cs.cfg_reference(), iq, n_epochs, L=L), ref_f64.run(ref_f64.plan_reference(), iq, n_epochs, L=L))


@pytest.mark.gpu
@settings(max_examples=150, deadline=None, derandomize=True, suppress_health_check=[HealthCheck.function_scoped_fixture])
@given(ref_traffic())
def test_gpu_reference_mode_on_random_traffic(built, t):
    iq, n_epochs, L = _ref_case(t)
    s = cs.Sensor(cs.cfg_reference())
    got = s.run_host(iq, n_epochs, L=L)
    s.close()
    _check_ref(got, ref_f64.run(ref_f64.plan_reference(), iq, n_epochs, L=L))
