#!/usr/bin/env python3
"""Which workgroups run fast: time of each workgroup's first epoch close against where it runs (XCD, SE, CU) and against its rank
among the workgroups that share its CU.  Variant 17 (A/B build): s_memrealtime at workgroup start and epoch close, HW_ID / XCC_ID."""
import os
import sys

os.environ.setdefault("CRN_SENSE_AB", "1")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "cognitive-radio-network_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import crnsense as cs  # noqa: E402

cfg = cs.cfg_energy_scaled(4096, 4.0)
spe = cs.samples_per_epoch(cfg)
dev = torch.device("cuda", 0)
s = cs.Sensor(cfg)
E = 28672
iq = torch.zeros(cs.samples_needed(cfg, E) * 2, dtype=torch.float32, device=dev)
stream = torch.cuda.current_stream().cuda_stream
s.synth_fill_device(iq.data_ptr(), E, spe, seed=1, stream=stream)
s.set_variant(17)
feats = torch.empty(E, cfg.n_bands, dtype=torch.float32, device=dev)
occ = torch.empty(E, cfg.n_bands, dtype=torch.uint8, device=dev)
dec = torch.empty(E, dtype=torch.int32, device=dev)
tr = torch.zeros(E * 4, dtype=torch.int64, device=dev)
outs = {"features": feats.data_ptr(), "ann_out": tr.data_ptr(), "decision": dec.data_ptr(), "occupancy": occ.data_ptr(), "spectrum": 0}
for _ in range(30):
    s.run_device(iq.data_ptr(), E, 4096, outs, stream=stream)
torch.cuda.synchronize()
a = tr.cpu().numpy()
closes = a[:E * 3].reshape(E, 3)[:, 0]
n_big, epw = (E - 1024) // 4, 4           # the default geometry of this batch: 6912 workgroups of 4 epochs, then 1024 of one
n_wg = n_big + 1024
start = a[E * 3:E * 3 + n_wg]
hw = a[E * 3 + E // 2:E * 3 + E // 2 + n_wg].view(np.uint64)
xcc = ((hw >> np.uint64(32)) & np.uint64(0xF)).astype(int)
hwid = (hw & np.uint64(0xFFFFFFFF)).astype(np.int64)
simd, cu, sh, se = (hwid >> 4) & 3, (hwid >> 8) & 15, (hwid >> 12) & 1, (hwid >> 13) & 3
first_epoch = np.concatenate([np.arange(n_big) * epw, n_big * epw + np.arange(1024)])
t0 = start.min()
dur = (closes[first_epoch] - start) / 100.0          # us from the workgroup's start to its first close
st = (start - t0) / 100.0
first = np.arange(n_wg) < 1024                        # the first round: all start together
print(f"first round (1024 workgroups, started within {st[first].max():.1f} us): start -> first close  min {dur[first].min():.1f}  median {np.median(dur[first]):.1f}  max {dur[first].max():.1f} us")
print("by XCD:      " + "  ".join(f"{x}: {np.median(dur[first & (xcc == x)]):.1f}" for x in sorted(set(xcc[first]))))
print("by SE:       " + "  ".join(f"{x}: {np.median(dur[first & (se == x)]):.1f}" for x in sorted(set(se[first]))))
place = xcc * 1000 + se * 100 + sh * 50 + cu         # one CU
rank = np.zeros(n_wg, dtype=int)
for pl in set(place[first]):
    idx = np.where(first & (place == pl))[0]
    rank[idx[np.argsort(start[idx], kind="stable")]] = np.arange(len(idx))
print("workgroups per CU in the first round: " + str(sorted(set(np.bincount(np.unique(place[first], return_inverse=True)[1])))))
print("by start order on its CU: " + "  ".join(f"#{r + 1}: {np.median(dur[first & (rank == r)]):.1f} (n={int((first & (rank == r)).sum())})" for r in range(int(rank[first].max()) + 1)))
print("by SIMD of wave 0:  " + "  ".join(f"{x}: {np.median(dur[first & (simd == x)]):.1f}" for x in sorted(set(simd[first]))))
# steady state: the big workgroups of the middle of the launch, time per epoch
mid = (np.arange(n_wg) >= 2048) & (np.arange(n_wg) < n_big - 1024)
per_epoch = (closes[np.minimum(first_epoch + 3, E - 1)] - closes[first_epoch]) / 3.0 / 100.0
print(f"middle of the launch, time per epoch of a workgroup: p10 {np.percentile(per_epoch[mid], 10):.1f}  median {np.median(per_epoch[mid]):.1f}  p90 {np.percentile(per_epoch[mid], 90):.1f} us;  by XCD: "
      + "  ".join(f"{x}: {np.median(per_epoch[mid & (xcc == x)]):.1f}" for x in sorted(set(xcc[mid]))))
# slot occupancy: a workgroup holds its slot from its start to its last close (+ the close itself, ~1.5 us)
last_epoch = np.concatenate([np.arange(n_big) * epw + epw - 1, n_big * epw + np.arange(1024)])
life = (closes[last_epoch] - start) / 100.0 + 1.5
span = (closes.max() - t0) / 100.0 + 1.5
print(f"slot occupancy: sum of workgroup lifetimes / (1024 slots x {span:.1f} us) = {life.sum() / (1024 * span):.3f}")
big_mid = mid
first_ep = dur[big_mid]
print(f"middle of the launch, 4-epoch workgroups: lifetime median {np.median(life[big_mid]):.1f} us; first epoch (start -> first close) median {np.median(first_ep):.1f} "
      f"(p10 {np.percentile(first_ep, 10):.1f}, p90 {np.percentile(first_ep, 90):.1f}); later epochs {np.median(per_epoch[big_mid]):.1f} us each")
# gap between a workgroup's end and the start of the next one on the same CU slot cannot be seen directly (the slot is not stamped);
# per CU: idle = 4 x span - sum of lifetimes there
cu_life = {}
for pl, l in zip(place, life):
    cu_life[pl] = cu_life.get(pl, 0.0) + l
vals = np.array(list(cu_life.values())) / (4 * span)
print(f"per CU: occupancy of its 4 slots  min {vals.min():.3f}  median {np.median(vals):.3f}  max {vals.max():.3f}  ({len(vals)} CUs)")
