#!/usr/bin/env python3
"""File-output counterpart of the reference's GNU Radio monitor (spectrum_analyzer.py:185-191,
262-275: uhd source -> qtgui.freq_sink_c(1024, Blackman-Harris, average 0.1) + waterfall), on the
MI355X sensing kernel.  No Qt, no radio: reads interleaved complex64 IQ from a file, writes one PSD
row per update (the waterfall) and the exponentially averaged trace.

  python tools/spectrum_monitor.py capture.c64 --fft 1024 --frames 8 --alpha 0.1 --out psd.npy

PSD row = 10 log10( mean over `frames` windowed FFTs of |X[k]|^2 / (N * sum w^2) ), fftshifted so
the centre frequency sits in the middle, like the freq sink.
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cognitive-radio-network_amd"))
import crnsense as cs  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("iq_file")
    ap.add_argument("--fft", type=int, default=1024)          # spectrum_analyzer.py:29
    ap.add_argument("--frames", type=int, default=8, help="FFT frames averaged per waterfall row")
    ap.add_argument("--alpha", type=float, default=0.1)       # spectrum_analyzer.py:270 set_fft_average(0.1)
    ap.add_argument("--out", default="psd.npy")
    ap.add_argument("--device", type=int, default=0)
    a = ap.parse_args()

    cfg = cs.cfg_energy_scaled(a.fft, 4.0)
    cfg.window, cfg.decide, cfg.frames_per_epoch, cfg.device = cs.WINDOW_BLACKMAN_HARRIS, cs.DECIDE_NONE, a.frames, a.device
    iq = np.fromfile(a.iq_file, dtype=np.float32)
    spe = cs.samples_per_epoch(cfg)
    n_rows = iq.size // (2 * spe)
    if n_rows < 1:
        raise SystemExit("capture shorter than one row")
    s = cs.Sensor(cfg)
    spec = s.run_host(iq[: n_rows * spe * 2], n_rows, want_spectrum=True)["spectrum"]
    s.close()
    n = np.arange(a.fft)
    x = 2 * np.pi * n / (a.fft - 1)
    w = 0.35875 - 0.48829 * np.cos(x) + 0.14128 * np.cos(2 * x) - 0.01168 * np.cos(3 * x)
    psd = np.fft.fftshift(spec, axes=1) / (a.fft * np.sum(w * w))
    avg = np.empty_like(psd)
    acc = psd[0].copy()
    for i in range(n_rows):  # single-pole IIR, as the freq sink's averaging
        acc = a.alpha * psd[i] + (1 - a.alpha) * acc
        avg[i] = acc
    np.save(a.out, {"waterfall_db": 10 * np.log10(np.maximum(psd, 1e-30)), "average_db": 10 * np.log10(np.maximum(avg, 1e-30))},
            allow_pickle=True)
    print(f"{n_rows} rows x {a.fft} bins -> {a.out}; peak bin of last averaged row: {int(avg[-1].argmax()) - a.fft // 2:+d}")


if __name__ == "__main__":
    main()
