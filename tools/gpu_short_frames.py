#!/usr/bin/env python3
"""Reference-mode kernel on the reference's real packet size (L = 364 of N = 512, zero-padded):
kernel time and rate per *input* sample.  Use with CRN_SENSE_LIB to A/B two builds."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "cognitive-radio-network_amd"))
import torch
import crnsense as cs

dev = torch.device("cuda", 0)
for L in (364, 100, 512):
    cfg = cs.cfg_reference()
    E = 229376
    spe = cs.samples_per_epoch(cfg, L)
    s = cs.Sensor(cfg)
    iq = torch.randn(cs.samples_needed(cfg, E, L) * 2, dtype=torch.float32, device=dev) * 1e-3
    feats = torch.empty(E, 4, dtype=torch.float32, device=dev)
    dec = torch.empty(E, dtype=torch.int32, device=dev)
    ann = torch.empty(E, 3, dtype=torch.float64, device=dev)
    occ = torch.empty(E, 4, dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    outs = {"features": feats.data_ptr(), "ann_out": ann.data_ptr(), "decision": dec.data_ptr(), "occupancy": occ.data_ptr(), "spectrum": 0}
    for _ in range(30):
        s.run_device(iq.data_ptr(), E, L, outs, stream=stream)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(40):
        s.run_device(iq.data_ptr(), E, L, outs, stream=stream)
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 40
    print(f"L={L}: {ms:.4f} ms per {E} epochs = {E * 10 / ms / 1e3:.0f} M frames/s, {E * spe * 8 / (ms * 1e-3) / 1e12:.2f} TB/s of input")
    s.close()
