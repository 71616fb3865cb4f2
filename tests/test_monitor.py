"""SURVEY.md §8(f)-4, the spectrum-analyser counterpart: tools/spectrum_monitor.py driven on a synthetic
capture, against an independent float64 numpy statement of what the reference's GNU Radio flowgraph draws
(spectrum_analyzer.py:262-275: freq_sink_c(1024, Blackman-Harris), set_fft_average(0.1), waterfall): window
multiply -> FFT -> 10 log10(|X / N|^2) -> fftshift -> single-pole IIR over updates."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _capture(n_rows, fft, frames, seed):
    rng = np.random.default_rng(seed)
    n = n_rows * frames * fft
    t = np.arange(n)
    x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)) * 1e-3
    x += 0.02 * np.exp(2j * np.pi * (100.3 / fft) * t)                         # an off-grid carrier (window leakage matters)
    x[n // 2:] += 0.05 * np.exp(2j * np.pi * (-200.0 / fft) * t[n // 2:])      # a second one that appears half way
    return x.astype(np.complex64)


def _reference(x, fft, frames, alpha, kind):
    i = np.arange(fft)
    th = 2 * np.pi * i / (fft - 1)
    w = (0.35875 - 0.48829 * np.cos(th) + 0.14128 * np.cos(2 * th) - 0.01168 * np.cos(3 * th)).astype(np.float32).astype(np.float64)
    rows = x.astype(np.complex128).reshape(-1, frames, fft)
    p = (np.abs(np.fft.fft(rows * w, axis=2)) ** 2).mean(axis=1)
    p = np.fft.fftshift(p, axes=1) / (fft ** 2 if kind == "gnuradio" else fft * np.sum(w * w))
    water = 10 * np.log10(np.maximum(p, 1e-30))
    v = water if kind == "gnuradio" else p
    avg = np.empty_like(v)
    acc = v[0].copy()
    for r in range(v.shape[0]):
        acc = v[r] if r == 0 else (1 - alpha) * acc + alpha * v[r]
        avg[r] = acc
    return water, (avg if kind == "gnuradio" else 10 * np.log10(np.maximum(avg, 1e-30)))


@pytest.mark.parametrize("kind,fft,frames,chunk", [("gnuradio", 1024, 1, 7), ("psd", 1024, 4, 4096), ("gnuradio", 4096, 2, 5)])
def test_spectrum_monitor_tool_matches_float64(built, tmp_path, kind, fft, frames, chunk):
    n_rows, alpha = 23, 0.1
    x = _capture(n_rows, fft, frames, seed=fft + frames)
    cap = tmp_path / "capture.c64"
    x.tofile(cap)
    out = tmp_path / "psd.npz"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "spectrum_monitor.py"), str(cap), "--fft", str(fft), "--frames",
                        str(frames), "--alpha", str(alpha), "--kind", kind, "--chunk-rows", str(chunk), "--out", str(out)],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    got = np.load(out)
    water, avg = _reference(x, fft, frames, alpha, kind)
    assert got["waterfall_db"].shape == (n_rows, fft)
    # A single fp32 frame leaves ~1e-7 of the peak AMPLITUDE on every bin: 1e-4 of a bin 60 dB down (9e-4 dB), 3e-3 of one 90 dB down
    # (0.03 dB).  Held: 2e-3 dB within 60 dB of the peak, 3e-2 dB within 90 dB, 0.1 dB on every bin.
    strong, deep = water > water.max() - 60, water > water.max() - 90
    assert np.abs(got["waterfall_db"] - water)[strong].max() < 2e-3 and np.abs(got["waterfall_db"] - water)[deep].max() < 3e-2
    assert np.abs(got["average_db"] - avg)[strong].max() < 2e-3 and np.abs(got["average_db"] - avg)[deep].max() < 3e-2
    assert np.abs(got["waterfall_db"] - water).max() < 0.1 and np.abs(got["average_db"] - avg).max() < 0.1
    # what the display shows: the carrier that appears half way rises with the IIR's time constant
    col = fft // 2 - 200
    assert got["average_db"][n_rows // 2 - 1, col] < got["average_db"][-1, col] - 10
    assert "peak bin of last averaged row" in r.stdout


def test_spectrum_monitor_tool_on_a_wire_format_capture(built, tmp_path):
    """--sc16: a capture of int16 pairs (what `uhd_rx_cfile --type short` writes) goes to the GPU as it is; the rows are the ones the
    float capture of the same samples gives, bit for bit.  (The tool loads libcrnsense_sc16.so for it: the optional build.)"""
    if not os.path.exists(os.path.join(ROOT, "cognitive-radio-network_amd", "libcrnsense_sc16.so")):
        pytest.skip("libcrnsense_sc16.so was not built (make -C csrc SC16=1)")
    fft, frames, n_rows = 1024, 2, 11
    x = _capture(n_rows, fft, frames, seed=9)
    raw = np.round(np.stack([x.real, x.imag], axis=1) * 32768.0 * 8).astype(np.int16)        # x 8: use a few more of the 16 bits
    (tmp_path / "cap.sc16").write_bytes(raw.tobytes())
    (raw.astype(np.float32) / np.float32(32768.0)).tofile(tmp_path / "cap.c64")
    outs = []
    for name, extra in (("cap.c64", []), ("cap.sc16", ["--sc16"])):
        out = tmp_path / (name + ".npz")
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "spectrum_monitor.py"), str(tmp_path / name), "--fft", str(fft),
                            "--frames", str(frames), "--out", str(out)] + extra, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(np.load(out))
    assert np.array_equal(outs[0]["waterfall_db"], outs[1]["waterfall_db"]) and np.array_equal(outs[0]["average_db"], outs[1]["average_db"])


def test_monitor_and_reserve_argument_errors(built):
    import ctypes as C
    import torch
    sys.path[:0] = [os.path.join(ROOT, "cognitive-radio-network_amd")]
    import crnsense as cs
    dev = torch.device("cuda", 0)
    cfg = cs.cfg_energy_scaled(1024, 4.0)
    cfg.window, cfg.decide = cs.WINDOW_BLACKMAN_HARRIS, cs.DECIDE_NONE
    s = cs.Sensor(cfg)
    spec = torch.zeros(4, 1024, dtype=torch.float32, device=dev)
    state = torch.zeros(1024, dtype=torch.float32, device=dev)
    for bad in (dict(kind=7), dict(alpha=0.0), dict(alpha=1.5), dict(n_rows=-1)):
        kw = dict(n_rows=4, kind=cs.MONITOR_GNURADIO, alpha=0.1)
        kw.update(bad)
        with pytest.raises(cs.CrnError):
            s.monitor_rows_device(spec.data_ptr(), kw["n_rows"], kw["kind"], kw["alpha"], True, state.data_ptr())
    with pytest.raises(cs.CrnError):
        s.monitor_rows_device(spec.data_ptr(), 4, cs.MONITOR_PSD, 0.1, True, 0)      # no state buffer
    s.monitor_rows_device(spec.data_ptr(), 0, cs.MONITOR_PSD, 0.1, True, state.data_ptr())   # zero rows: nothing to do
    with pytest.raises(cs.CrnError):
        s.reserve_host(0)
    s.reserve_host(3, want_spectrum=True)
    ring = cs.Ingest(s, 1, 100, 1)
    with pytest.raises(cs.CrnError):
        ring.set_packet_len(101)       # longer than the ring was sized for
    ring.set_packet_len(64)
    ring.close()
    s.close()
