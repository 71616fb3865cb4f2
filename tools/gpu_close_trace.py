#!/usr/bin/env python3
"""Time the epoch close from inside the kernel (variant 17: s_memtime stamps at close entry/exit)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "cognitive-radio-network_amd"))
import numpy as np
import torch
import os as _os
_os.environ.setdefault("CRN_SENSE_AB", "1")   # measurement variants: libcrnsense_ab.so
import crnsense as cs

cfg = cs.cfg_energy_scaled(4096, 4.0)
spe = cs.samples_per_epoch(cfg)
E = 28672
dev = torch.device("cuda", 0)
s = cs.Sensor(cfg)
iq = torch.zeros(cs.samples_needed(cfg, E) * 2, dtype=torch.float32, device=dev)
feats = torch.empty(E, cfg.n_bands, dtype=torch.float32, device=dev)
occ = torch.empty(E, cfg.n_bands, dtype=torch.uint8, device=dev)
dec = torch.empty(E, dtype=torch.int32, device=dev)
tr4 = torch.zeros(E * 4, dtype=torch.int64, device=dev)   # [E][3] close stamps + one start stamp per workgroup behind them
tr = tr4[:E * 3].view(E, 3)
stream = torch.cuda.current_stream().cuda_stream
s.synth_fill_device(iq.data_ptr(), E, spe, seed=1, stream=stream)
outs = {"features": feats.data_ptr(), "ann_out": tr4.data_ptr(), "decision": dec.data_ptr(), "occupancy": occ.data_ptr(), "spectrum": 0}
s.set_variant(17)
for _ in range(40):
    s.run_device(iq.data_ptr(), E, 4096, outs, stream=stream)
torch.cuda.synchronize()
a = tr.cpu().numpy().astype(np.int64)
enter0, packed, enter3 = a[:, 0], a[:, 1].view(np.uint64), a[:, 2]
d = [((packed >> np.uint64(16 * i)) & np.uint64(0xFFFF)).astype(np.int64) for i in range(4)]
for name, x in (("band sums done", d[0]), ("barrier passed", d[1]), ("features ready", d[2]), ("exit", d[3])):
    print("  +%-16s median %5d  p10 %5d  p90 %5d" % (name, np.median(x), np.percentile(x, 10), np.percentile(x, 90)))
dur = d[3]
skew = enter3 - enter0
print("close duration of the first wave [memtime ticks]: median %d  p10 %d  p90 %d  mean %.0f" % (np.median(dur), np.percentile(dur, 10), np.percentile(dur, 90), dur.mean()))
print("entry skew last wave - first wave [10 ns]: median %d  p10 %d  p90 %d" % (np.median(skew), np.percentile(skew, 10), np.percentile(skew, 90)))
# epochs 4g..4g+3 belong to one workgroup (groups_per_wg = 4): time between consecutive closes
e = enter0.reshape(-1, 4)
gap = (e[:, 1:] - e[:, :-1]).ravel()
print("entry-to-entry of consecutive epochs in a workgroup (10 frames + close) [10 ns]: median %d  p10 %d p90 %d" % (np.median(gap), np.percentile(gap, 10), np.percentile(gap, 90)))
print("whole kernel span in ticks: %d" % (a[:, 0].max() - a[:, 0].min()))

# workgroup occupancy of the machine: a workgroup (4 consecutive epochs) is busy from about one epoch
# before its first close to its last close; 1024 workgroup slots (256 CUs x 4)
first, last = e[:, 0], e[:, 3]
per_epoch = (e[:, 3] - e[:, 0]) / 3.0
start = first - per_epoch
span = last.max() - start.min()
busy = (last - start).sum()
print("workgroups %d, span %.1f us, slot utilisation %.3f" % (e.shape[0], span / 100.0, busy / (1024.0 * span)))
order = np.sort(last)
print("finish times of the last 1024 workgroups relative to the end [10 ns]: p50 %d  p10 %d  min %d" % (
    np.median(order[-1024:]) - order[-1], np.percentile(order[-1024:], 10) - order[-1], order[-1024] - order[-1]))
print("start times of the first 1024 workgroups relative to the start [10 ns]: p50 %d  p90 %d  max %d" % (
    np.median(np.sort(start)[:1024]) - start.min(), np.percentile(np.sort(start)[:1024], 90) - start.min(), np.sort(start)[1023] - start.min()))
