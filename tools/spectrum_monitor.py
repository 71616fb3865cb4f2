#!/usr/bin/env python3
"""File-output counterpart of the reference's GNU Radio monitor (spectrum_analyzer.py:185-191,
262-275: uhd source -> qtgui.freq_sink_c(1024, Blackman-Harris, average 0.1) + waterfall), on the
MI355X: the windowed FFT + |X|^2 is the sensing kernel, the dB conversion, fftshift and the
exponential averaging are crn_monitor_rows_device — the capture is uploaded once and only the
drawn rows come back.  No Qt, no radio: reads interleaved complex64 IQ from a file, writes one dB
row per update (the waterfall) and the averaged trace.

  python tools/spectrum_monitor.py capture.c64 --fft 1024 --frames 1 --alpha 0.1 --out psd.npz

--kind gnuradio (default): 10 log10(|X / N|^2), IIR on the dB values — gr-qtgui's freq sink as published
--kind psd:                10 log10(mean |X|^2 / (N sum w^2)), IIR on linear power
Rows are processed in chunks, the IIR state staying on the device between chunks.
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cognitive-radio-network_amd"))
if "--sc16" in sys.argv and "CRN_SENSE_LIB" not in os.environ:
    # captures of int16 pairs need the optional wire-format kernels: libcrnsense_sc16.so (make -C csrc SC16=1), chosen before the binding loads
    os.environ["CRN_SENSE_LIB"] = os.path.join(ROOT, "cognitive-radio-network_amd", "libcrnsense_sc16.so")
import crnsense as cs  # noqa: E402


def run(iq_file, fft=1024, frames=1, alpha=0.1, kind="gnuradio", device=0, chunk_rows=4096, sc16=False):
    import torch
    cfg = cs.cfg_energy_scaled(fft, 4.0)
    cfg.window, cfg.decide, cfg.frames_per_epoch, cfg.device = cs.WINDOW_BLACKMAN_HARRIS, cs.DECIDE_NONE, frames, device
    # sc16: a capture in the radio's wire format (int16 pairs, e.g. `uhd_rx_cfile --type short`): it stays int16 in HBM
    iq = np.fromfile(iq_file, dtype=np.int16 if sc16 else np.float32)
    spe = cs.samples_per_epoch(cfg)
    n_rows = iq.size // (2 * spe)
    if n_rows < 1:
        raise SystemExit("capture shorter than one row")
    dev = torch.device("cuda", device)
    torch.cuda.set_device(device)
    stream = torch.cuda.current_stream().cuda_stream
    s = cs.Sensor(cfg)
    d_iq = torch.from_numpy(iq[: n_rows * spe * 2]).to(dev)
    state = torch.zeros(fft, dtype=torch.float32, device=dev)
    water = torch.empty(n_rows, fft, dtype=torch.float32, device=dev)
    avg = torch.empty(n_rows, fft, dtype=torch.float32, device=dev)
    spec = torch.empty(min(chunk_rows, n_rows), fft, dtype=torch.float32, device=dev)
    k = cs.MONITOR_GNURADIO if kind == "gnuradio" else cs.MONITOR_PSD
    for r0 in range(0, n_rows, chunk_rows):
        n = min(chunk_rows, n_rows - r0)
        s.run_device(d_iq.data_ptr() + r0 * spe * (4 if sc16 else 8), n, fft, {"features": 0, "ann_out": 0, "decision": 0, "occupancy": 0,
                                                                             "spectrum": spec.data_ptr()}, stream=stream, sc16=sc16)
        s.monitor_rows_device(spec.data_ptr(), n, k, alpha, r0 == 0, state.data_ptr(),
                              water.data_ptr() + r0 * fft * 4, avg.data_ptr() + r0 * fft * 4, stream=stream)
    torch.cuda.synchronize()
    s.close()
    return water.cpu().numpy(), avg.cpu().numpy()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("iq_file")
    ap.add_argument("--fft", type=int, default=1024)          # spectrum_analyzer.py:29
    ap.add_argument("--frames", type=int, default=1, help="FFT frames averaged per row (the freq sink draws one FFT per update)")
    ap.add_argument("--alpha", type=float, default=0.1)       # spectrum_analyzer.py:270 set_fft_average(0.1)
    ap.add_argument("--kind", choices=["gnuradio", "psd"], default="gnuradio")
    ap.add_argument("--chunk-rows", type=int, default=4096)
    ap.add_argument("--out", default="psd.npz")
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--sc16", action="store_true", help="the capture holds int16 pairs (the radio's wire format) instead of complex64")
    a = ap.parse_args()
    water, avg = run(a.iq_file, a.fft, a.frames, a.alpha, a.kind, a.device, a.chunk_rows, a.sc16)
    np.savez(a.out, waterfall_db=water, average_db=avg)
    print(f"{water.shape[0]} rows x {a.fft} bins -> {a.out}; peak bin of last averaged row: {int(avg[-1].argmax()) - a.fft // 2:+d}")


if __name__ == "__main__":
    main()
