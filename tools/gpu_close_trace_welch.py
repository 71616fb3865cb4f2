#!/usr/bin/env python3
"""Epoch-close timing of the Welch kernel (LDS form of the close, 64 bands): variant 17 stamps."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "cognitive-radio-network_amd"))
import numpy as np
import torch
import os as _os
_os.environ.setdefault("CRN_SENSE_AB", "1")   # measurement variants: libcrnsense_ab.so
import crnsense as cs

cfg = cs.cfg_welch(4096, 8, 64)
for b in range(64):
    cfg.thresh[b] = 1e-2
spe = cs.samples_per_epoch(cfg)
E = 71680
dev = torch.device("cuda", 0)
s = cs.Sensor(cfg)
iq = torch.zeros(cs.samples_needed(cfg, E) * 2, dtype=torch.float32, device=dev)
feats = torch.empty(E, 64, dtype=torch.float32, device=dev)
occ = torch.empty(E, 64, dtype=torch.uint8, device=dev)
dec = torch.empty(E, dtype=torch.int32, device=dev)
tr4 = torch.zeros(E * 4, dtype=torch.int64, device=dev)   # [E][3] close stamps + one start stamp per workgroup behind them
tr = tr4[:E * 3].view(E, 3)
stream = torch.cuda.current_stream().cuda_stream
s.synth_fill_device(iq.data_ptr(), E, spe, seed=1, stream=stream)
outs = {"features": feats.data_ptr(), "ann_out": tr4.data_ptr(), "decision": dec.data_ptr(), "occupancy": occ.data_ptr(), "spectrum": 0}
s.set_variant(17)
for _ in range(20):
    s.run_device(iq.data_ptr(), E, 4096, outs, stream=stream)
torch.cuda.synchronize()
a = tr.cpu().numpy().astype(np.int64)
enter0, packed = a[:, 0], a[:, 1].view(np.uint64)
d = [((packed >> np.uint64(16 * i)) & np.uint64(0xFFFF)).astype(np.int64) for i in range(4)]
for name, x in (("spectrum image visible", d[0]), ("band sums done (first wave)", d[1]), ("features ready", d[2]), ("exit", d[3])):
    print("  +%-28s median %5d  p10 %5d  p90 %5d  [s_memtime ticks]" % (name, np.median(x), np.percentile(x, 10), np.percentile(x, 90)))
e = enter0.reshape(-1, 4)
gap = (e[:, 1:] - e[:, :-1]).ravel()
print("entry-to-entry of consecutive epochs in a workgroup (8 frames + close) [10 ns]: median %d p10 %d p90 %d" % (np.median(gap), np.percentile(gap, 10), np.percentile(gap, 90)))
