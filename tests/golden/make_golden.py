#!/usr/bin/env python3
"""Regenerates the fixtures in tests/golden/.

What these vectors are — and are not.  The reference ships no test vectors for the sensing path
and cannot be built in this image (no liquid-dsp / UHD), so there is nothing of the reference's to
record: PARITY IS UNPINNED.  The expected outputs here come from this repo's own CPU oracle
(oracle/crn_oracle.c, a restatement of CE_Predictive_Node.cpp:146-289) plus a float64 numpy DFT,
and the known-answer entries are derived by hand from the DFT definition.  They pin the oracle
against regressions and give the GPU tests inputs that do not depend on a random generator's
version.

  python tests/golden/make_golden.py        (from the repo root, after __graft_entry__.build())
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [os.path.join(ROOT, "cognitive-radio-network_amd"), os.path.join(ROOT, "tests")]

import crnsense as cs  # noqa: E402
import oracle_py as orc  # noqa: E402
import signals  # noqa: E402


def epochs_fixture(cfg, n_epochs, seed, L, name):
    iq, picks = signals.make_epochs(cfg, n_epochs, seed=seed, L=L)
    want = orc.run(cfg, iq, n_epochs, L=L, want_spectrum=True)
    truth = signals.spectrum_f64(cfg, iq, n_epochs, L=L)
    np.savez_compressed(os.path.join(HERE, name), iq=iq, picks=picks.astype(np.int32), L=np.int32(L),
                        spectrum=want["spectrum"], spectrum_f64=truth.astype(np.float64),
                        features=want["features"], ann_out=want["ann_out"], decision=want["decision"],
                        occupancy=want["occupancy"])


def ann_fixture():
    rng = np.random.default_rng(2024)
    feats = (10 ** rng.uniform(-3, 3, size=(128, 4))).astype(np.float32)
    feats[:, 0] = (10 ** rng.uniform(-3, 1.5, size=128)).astype(np.float32)  # NF in the calibrated range
    feats[0] = 0
    outs = np.zeros((128, 3))
    dec = np.zeros(128, np.int32)
    for i in range(128):
        dec[i], outs[i] = orc.ann(feats[i])
    keep = (np.abs(outs - 0.8) > 1e-3).all(axis=1)
    np.savez_compressed(os.path.join(HERE, "ann_table.npz"), features=feats[keep], ann_out=outs[keep], decision=dec[keep])


def kat_fixture():
    """Hand-derived answers (no oracle involved): a unit tone on bin b of a 512-point frame gives
    |X[b]| = 512 and 0 elsewhere; ten such frames average to fft_avg[b] = 512; the band holding b
    sums to 512 and its feature is 512^2 = 262144, all other features are 0."""
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
    zero = {"features": [0, 0, 0, 0], "ann_out": [0.4790, 4.12e-5, 3.35e-3], "decision": 0}
    json.dump({"n": 512, "frames": 10, "tone_rows": rows, "all_zero": zero}, open(os.path.join(HERE, "kat.json"), "w"),
              indent=1)


if __name__ == "__main__":
    epochs_fixture(cs.cfg_reference(), 8, 0xC0FFEE, 364, "ref512_L364.npz")
    epochs_fixture(cs.cfg_energy_scaled(1024, 4.0), 3, 0xC0FFEE + 1, 1024, "energy1024.npz")
    epochs_fixture(cs.cfg_energy_scaled(4096, 4.0), 1, 0xC0FFEE + 2, 4096, "energy4096.npz")
    ann_fixture()
    kat_fixture()
    print(sorted(os.listdir(HERE)))
