#!/usr/bin/env python3
"""Welch stream (cfg2: 4096-pt Hann, hop N/2, 64 bands, K = 8): kernel time against the length of the workgroup spans.  A span of E
epochs reads E K + 1 half-frames for E K new ones (its first half-frame is also the previous span's last), so traffic over
algorithmic bytes = 1 + workgroups / (epochs K).  crn_sense_set_variant 100 + E = epochs per big workgroup, 200 + n = n x 256 epochs
in tail workgroups, 300 + e = epochs per tail workgroup.  Interleaved repetitions on one box."""
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
grid = [(0, -1, 0)] + [(e, t, te) for e in (8, 16, 32, 48, 64) for t, te in ((3, 1), (3, 4), (6, 4), (12, 8), (24, 8))]
sensors = []
for e, t, te in grid:
    s = cs.Sensor(cfg)
    if e:
        s.set_variant(100 + e)
        s.set_variant(200 + t)
        s.set_variant(300 + te)
    sensors.append(s)
ms = {g: [] for g in grid}
for rep in range(4):
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
        assert torch.equal(feats, ref[0]) and torch.equal(occ, ref[1]), g      # the geometry changes nothing but the order of work
print(f"K = {K}, {E} epochs, {E * spe * 8 / 2**30:.2f} GiB per launch; big span E epochs, tail T x 256 epochs in spans of e")
for g in grid:
    e, t, te = g
    if e == 0:
        wgs, label = None, "default geometry"
    else:
        tail = min(t * 256, E // 4)
        big = (E - tail) // e
        wgs = big + -(-(E - big * e) // te)
        label = f"E={e:2d} T={t:2d} e={te}"
    m = float(np.mean(ms[g]))
    extra = "" if wgs is None else f"  workgroups {wgs:5d}  traffic x{1 + wgs / (E * K):.4f}"
    print(f"{label:20s} {m:.4f} ms  {E * spe * 8 / (m * 1e-3) / 8e12:.4f} of HBM peak  (min {min(ms[g]):.4f}){extra}")
