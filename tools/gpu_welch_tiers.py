#!/usr/bin/env python3
"""Welch stream (cfg2), three-tier launch: big spans E1, a middle tier of M x 256 epochs in spans of E2, a tail of T x 256 epochs in
spans of E3 (crn_sense_set_variant 100 + E1, 400 + E2, 500 + M, 300 + E3, 200 + T).  Interleaved repetitions on one box."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "cognitive-radio-network_amd")]
import crnsense as cs  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 8
cfg = cs.cfg_welch(4096, K, 64)
for b in range(64):
    cfg.thresh[b] = 4.0 * 64 * 4096 * 1e-6 * 0.375
spe = cs.samples_per_epoch(cfg)
E = (28672 * 40960) // spe
n = cs.samples_needed(cfg, E)
dev = torch.device("cuda", 0)
iq = torch.zeros(n * 2, dtype=torch.float32, device=dev)
s0 = cs.Sensor(cfg)
s0.synth_fill_device(iq.data_ptr(), E, spe, seed=0xC0FFEE)
feats = torch.empty(E, 64, device=dev)
occ = torch.empty(E, 64, dtype=torch.uint8, device=dev)
dec = torch.empty(E, dtype=torch.int32, device=dev)
outs = {"features": feats.data_ptr(), "ann_out": 0, "decision": dec.data_ptr(), "occupancy": occ.data_ptr(), "spectrum": 0}
stream = torch.cuda.current_stream().cuda_stream
s0.run_device(iq.data_ptr(), E, 4096, outs, stream=stream)
torch.cuda.synchronize()
ref = (feats.clone(), occ.clone())
# (E1, E2, M, E3, T): M, T in units of 256 epochs; E2 = 0: two tiers
grid = [None, (32, 0, 0, 8, 24), (32, 8, 24, 1, 3), (32, 8, 24, 2, 6), (32, 8, 18, 2, 6), (40, 8, 24, 2, 6), (32, 8, 24, 1, 6), (32, 12, 36, 2, 6),
        (24, 8, 24, 2, 6), (32, 8, 30, 4, 6), (48, 12, 36, 2, 6), (32, 16, 48, 4, 12), (28, 8, 24, 2, 3)]
sensors = []
for g in grid:
    s = cs.Sensor(cfg)
    if g:
        e1, e2, m, e3, t = g
        s.set_variant(100 + e1)
        s.set_variant(400 + e2)
        s.set_variant(500 + m)
        s.set_variant(300 + e3)
        s.set_variant(200 + t)
    sensors.append(s)
ms = {g: [] for g in grid}
for rep in range(5):
    for g, s in zip(grid, sensors):
        for _ in range(6):
            s.run_device(iq.data_ptr(), E, 4096, outs, stream=stream)
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(12)]
        for a, b in ev:
            a.record()
            s.run_device(iq.data_ptr(), E, 4096, outs, stream=stream)
            b.record()
        torch.cuda.synchronize()
        ms[g] += [a.elapsed_time(b) for a, b in ev]
        assert torch.equal(feats, ref[0]) and torch.equal(occ, ref[1]), g
print(f"K = {K}, {E} epochs; spans: big E1, middle E2 over M x 256 epochs, tail E3 over T x 256 epochs")
for g in grid:
    m = float(np.mean(ms[g]))
    if g is None:
        label, extra = "default geometry", ""
    else:
        e1, e2, mm, e3, t = g
        tail, mid = t * 256, (mm * 256 if e2 else 0)
        big = (E - tail - mid) // e1
        wgs = big + (mid // e2 if e2 else 0) + -(-(E - big * e1 - (mid // e2 * e2 if e2 else 0)) // e3)
        label = f"E1={e1} E2={e2} M={mm} E3={e3} T={t}"
        extra = f"  workgroups {wgs:5d}  reads x{1 + wgs / (E * K):.4f}"
    print(f"{label:32s} {m:.4f} ms  {E * spe * 8 / (m * 1e-3) / 8e12:.4f} of HBM peak  (min {min(ms[g]):.4f}){extra}")
