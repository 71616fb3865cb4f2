#!/usr/bin/env python3
"""Regenerates the fixtures in tests/golden/.

What these vectors are — and are not.  The reference ships no test vectors for the sensing path
and cannot be built in this image (no liquid-dsp / UHD), so there is nothing of the reference's to
record: PARITY IS UNPINNED against reference output.  What is recorded here instead comes from
tests/ref_f64.py — an independent float64 numpy statement of SURVEY.md Appendix A (spectrum -> band
sums -> square -> 4-5-3 network -> cascade) that shares no code with oracle/ or with the product —
so the C oracle (tests/test_golden.py, CPU) and the HIP path (-m gpu) are both compared with
numbers that neither of them produced.  The known-answer entries (kat.json) are derived by hand
from the DFT definition.

Inputs are seeded synthetic IQ (tests/signals.py: SURVEY.md §8(d) recipe).  Every epoch is asserted
to sit outside the margin band (|O - 0.8| > 1e-3, |F / thr - 1| > 1e-4) so that fp32
implementations can be required to reproduce the decisions exactly.

  python tests/golden/make_golden.py        (from the repo root, after __graft_entry__.build())
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [os.path.join(ROOT, "cognitive-radio-network_amd"), os.path.join(ROOT, "tests")]

import crnsense as cs  # noqa: E402  (configuration structs for the input generator only)
import ref_f64  # noqa: E402
import signals  # noqa: E402


def epochs_fixture(cfg, plan, n_epochs, seed, L, name, thresh=None):
    iq, picks = signals.make_epochs(cfg, n_epochs, seed=seed, L=L)
    want = ref_f64.run(plan, iq, n_epochs, L=L)
    if plan.decide == "ann":
        assert (want["margin"] > 1e-3).all(), (name, want["margin"])
        assert (want["decision"] == picks).all(), (name, want["decision"], picks)
    elif plan.decide == "threshold":
        assert (want["margin"] > 1e-4).all(), (name, want["margin"])
    extra = {} if thresh is None else {"thresh": np.asarray(thresh, np.float32)}
    np.savez_compressed(os.path.join(HERE, name), iq=iq, picks=picks.astype(np.int32), L=np.int32(L),
                        spectrum_f64=want["spectrum"], features_f64=want["features"], ann_out_f64=want["ann_out"],
                        decision=want["decision"], occupancy=want["occupancy"], margin=want["margin"], **extra)


def welch_fixture(name, n=1024, k=8, n_bands=64, n_epochs=2, seed=0xC0FFEE + 3, lam=4.0):
    """BASELINE.json configs[2] in small: Hann, 50 % overlap, 64 bands, thr_b = lam * median band energy
    (SURVEY.md §8(d) cfg2), thresholds stored as the f32 array both implementations are handed."""
    cfg = cs.cfg_welch(n, k, n_bands)
    iq, _ = signals.make_epochs(cfg, n_epochs, seed=seed, L=n)
    probe = ref_f64.run(ref_f64.plan_welch(n, k, n_bands, [np.inf] * n_bands), iq, n_epochs, L=n)
    thr = np.full(n_bands, lam * np.median(probe["features"]), dtype=np.float32)
    for b in range(n_bands):
        cfg.thresh[b] = float(thr[b])
    epochs_fixture(cfg, ref_f64.plan_welch(n, k, n_bands, thr), n_epochs, seed, n, name, thresh=thr)


def ann_fixture():
    rng = np.random.default_rng(2024)
    feats = (10 ** rng.uniform(-3, 3, size=(128, 4))).astype(np.float32)
    feats[:, 0] = (10 ** rng.uniform(-3, 1.5, size=128)).astype(np.float32)  # NF in the calibrated range
    feats[0] = 0
    # SURVEY.md Appendix C operating points: idle, and each channel at the occupied level
    feats[1] = (0.04, 0.39, 0.39, 0.39)
    for c in (1, 2, 3):
        feats[1 + c] = (0.04, 0.39, 0.39, 0.39)
        feats[1 + c, c] = 866.0
    outs = ref_f64.ann(feats)
    dec = ref_f64.cascade(outs)
    keep = (np.abs(outs - 0.8) > 1e-3).all(axis=1)
    np.savez_compressed(os.path.join(HERE, "ann_table.npz"), features=feats[keep], ann_out=outs[keep], decision=dec[keep])


def kat_fixture():
    """Hand-derived answers (no implementation involved): a unit tone on bin b of a 512-point frame
    gives |X[b]| = 512 and 0 elsewhere; ten such frames average to fft_avg[b] = 512; the band holding
    b sums to 512 and its feature is 512^2 = 262144, all other features are 0."""
    edges = {"NF": (300, 310), "CH2": (55, 85), "CH3": (189, 222), "CH1a": (0, 16), "CH1b": (496, 511)}
    band = {"NF": 0, "CH1a": 1, "CH1b": 1, "CH2": 2, "CH3": 3}
    rows = []
    for name, (lo, hi) in edges.items():
        for b, inside in ((lo - 1, False), (lo, True), (hi - 1, True), (hi, False)):
            rows.append({"bin": int(b % 512), "band": band[name] if inside else None,
                         "feature": 262144.0 if inside else 0.0})
    # the wrap-around neighbours of CH1 are themselves inside other CH1 runs: fix by hand
    for r in rows:
        if r["bin"] == 511:
            r.update(band=None, feature=0.0)  # bin 511 is in no band (CE_Predictive_Node.cpp:177)
        if r["bin"] == 495:
            r.update(band=None, feature=0.0)
    # SURVEY.md Appendix C (survey-time numpy probe, typed from the document, not recomputed here)
    zero = {"features": [0, 0, 0, 0], "ann_out": [0.4790, 4.12e-5, 3.35e-3], "decision": 0}
    appendix_c = {"idle": {"features": [0.04, 0.39, 0.39, 0.39], "ann_out": [0.062, 1.0e-4, 1.2e-3]},
                  "crossing": {"1": 231.0, "2": 25.0, "3": 45.0},
                  "occupied_feature": 866.0, "occupied_output_range": [0.9993, 0.9995], "others_below": 3e-4}
    json.dump({"n": 512, "frames": 10, "tone_rows": rows, "all_zero": zero, "appendix_c": appendix_c},
              open(os.path.join(HERE, "kat.json"), "w"), indent=1)


if __name__ == "__main__":
    epochs_fixture(cs.cfg_reference(), ref_f64.plan_reference(), 8, 0xC0FFEE, 364, "ref512_L364.npz")
    epochs_fixture(cs.cfg_reference(), ref_f64.plan_reference(), 4, 0xC0FFEE + 4, 512, "ref512_L512.npz")
    epochs_fixture(cs.cfg_energy_scaled(1024, 4.0), ref_f64.plan_energy_scaled(1024, 4.0), 3, 0xC0FFEE + 1, 1024, "energy1024.npz")
    epochs_fixture(cs.cfg_energy_scaled(4096, 4.0), ref_f64.plan_energy_scaled(4096, 4.0), 1, 0xC0FFEE + 2, 4096, "energy4096.npz")
    welch_fixture("welch1024.npz")
    ann_fixture()
    kat_fixture()
    print(sorted(os.listdir(HERE)))
