#!/usr/bin/env python3
"""What the unfused route costs: the same 4096-point x 3-band workload as library calls — rocFFT
(torch.fft.fft) writes the complex spectrum to HBM, a second pass forms |X|^2 and the K-frame mean, a
third the band sums — next to the fused kernel on the same batch.  Informational (the product never
links rocFFT)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "cognitive-radio-network_amd"))
import torch
import crnsense as cs

dev = torch.device("cuda", 0)
cfg = cs.cfg_energy_scaled(4096, 4.0)
N, K = 4096, cfg.frames_per_epoch
E = 7168
spe = cs.samples_per_epoch(cfg)
s = cs.Sensor(cfg)
iq = torch.zeros(E * spe * 2, dtype=torch.float32, device=dev)
stream = torch.cuda.current_stream().cuda_stream
s.synth_fill_device(iq.data_ptr(), E, spe, seed=1, stream=stream)
feats = torch.empty(E, 4, dtype=torch.float32, device=dev)
occ = torch.empty(E, 4, dtype=torch.uint8, device=dev)
outs = {"features": feats.data_ptr(), "ann_out": 0, "decision": 0, "occupancy": occ.data_ptr(), "spectrum": 0}
bands = []
for b in range(cfg.n_bands):
    idx = torch.cat([torch.arange(cfg.segs[i].lo, cfg.segs[i].hi) for i in range(cfg.n_segs) if cfg.segs[i].band == b]).to(dev)
    bands.append(idx)
x = torch.view_as_complex(iq.view(E, K, N, 2))


def library_route():
    X = torch.fft.fft(x, dim=2)                       # rocFFT: reads 8 B, writes 8 B per sample
    P = (X.real * X.real + X.imag * X.imag).mean(dim=1)   # reads 8 B per sample
    return torch.stack([P[:, i].sum(dim=1) for i in bands], dim=1)


def timed(fn, n):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        r = fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n, r


t_lib, f_lib = timed(library_route, 10)
t_fused, _ = timed(lambda: s.run_device(iq.data_ptr(), E, N, outs, stream=stream), 40)
rel = ((f_lib - feats).abs() / feats.abs().clamp_min(1e-30)).max().item()
gs = E * spe / 1e9
print(f"batch {E} epochs = {E * spe * 8 / 2**30:.2f} GiB")
print(f"rocFFT + |X|^2 mean + band sums (torch): {t_lib:.3f} ms = {gs / (t_lib * 1e-3):.0f} Gsamples/s")
print(f"fused sensing kernel                   : {t_fused:.3f} ms = {gs / (t_fused * 1e-3):.0f} Gsamples/s   ({t_lib / t_fused:.1f}x)")
print(f"max relative difference of the band features between the two routes: {rel:.2e}")
