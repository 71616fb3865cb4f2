#!/usr/bin/env python3
"""SURVEY.md §8(d)'s 2 GiB batch (6553 epochs of 10 x 4096) launched round-robin on 1 / 2 / 3 / 4 streams: span of 64 launches,
interleaved repetitions on one box, outputs compared.  What one stream loses is its partly filled last round of workgroups."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "cognitive-radio-network_amd")]
import crnsense as cs  # noqa: E402

dev = torch.device("cuda", 0)
cfg = cs.cfg_energy_scaled(4096, 4.0)
spe = 40960
E_big, E = 28672, int(sys.argv[1]) if len(sys.argv) > 1 else 6553
iq = torch.zeros(E_big * spe * 2, dtype=torch.float32, device=dev)
s = cs.Sensor(cfg)
s.synth_fill_device(iq.data_ptr(), E_big, spe, seed=0xC0FFEE)
sets = []
for _ in range(8):
    t = [torch.empty(E, 4, device=dev), torch.empty(E, dtype=torch.int32, device=dev), torch.empty(E, 4, dtype=torch.uint8, device=dev)]
    sets.append((t, {"features": t[0].data_ptr(), "ann_out": 0, "decision": t[1].data_ptr(), "occupancy": t[2].data_ptr(), "spectrum": 0}))
streams = [torch.cuda.Stream(device=dev) for _ in range(4)]
n = 64
res = {k: [] for k in (1, 2, 3, 4)}
for rep in range(5):
    for k in (1, 2, 3, 4):
        torch.cuda.synchronize()
        for i in range(24):
            s.run_device(iq.data_ptr(), E, 4096, sets[i & 7][1], stream=streams[i % k].cuda_stream)
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True)
        ends = [torch.cuda.Event(enable_timing=True) for _ in range(k)]
        e0.record(streams[0])
        for st in streams[1:k]:
            st.wait_event(e0)
        for i in range(n):
            s.run_device(iq.data_ptr(), E, 4096, sets[i & 7][1], stream=streams[i % k].cuda_stream)
        for j in range(k):
            ends[j].record(streams[j])
        torch.cuda.synchronize()
        res[k].append(max(e0.elapsed_time(e) for e in ends) / n)
assert all(torch.equal(a, b) for a, b in zip(sets[0][0], sets[5][0]))
for k in (1, 2, 3, 4):
    m = float(np.mean(res[k]))
    print(f"{k} stream(s): {m:.4f} ms per launch = {E * spe * 8 / (m * 1e-3) / 8e12:.4f} of the HBM peak   (repetitions: " + " ".join(f"{x:.4f}" for x in res[k]) + ")")
