#!/usr/bin/env python3
"""Sweep the number of epoch groups handed to single-group workgroups at the end of the launch
(crn_sense_set_variant 200 + n = n x 256 groups).  Interleaved repetitions on one box."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "cognitive-radio-network_amd"))
import torch
import crnsense as cs

fft = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
cfg = cs.cfg_energy_scaled(fft, 4.0)
spe = cs.samples_per_epoch(cfg)
E = (28672 * 40960) // spe
dev = torch.device("cuda", 0)
s = cs.Sensor(cfg)
iq = torch.zeros(cs.samples_needed(cfg, E) * 2, dtype=torch.float32, device=dev)
truth = torch.empty(E, dtype=torch.int32, device=dev)
feats = torch.empty(E, cfg.n_bands, dtype=torch.float32, device=dev)
occ = torch.empty(E, cfg.n_bands, dtype=torch.uint8, device=dev)
dec = torch.empty(E, dtype=torch.int32, device=dev)
stream = torch.cuda.current_stream().cuda_stream
s.synth_fill_device(iq.data_ptr(), E, spe, seed=1, truth_ptr=truth.data_ptr(), stream=stream)
outs = {"features": feats.data_ptr(), "ann_out": 0, "decision": dec.data_ptr(), "occupancy": occ.data_ptr(), "spectrum": 0}
for _ in range(60):
    s.run_device(iq.data_ptr(), E, fft, outs, stream=stream)
torch.cuda.synchronize()
res = {}
for rep in range(3):
    for n in (0, 2, 4, 8, 12, 16, 24, 32):
        s.set_variant(200 + n)
        for _ in range(5):
            s.run_device(iq.data_ptr(), E, fft, outs, stream=stream)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(40):
            s.run_device(iq.data_ptr(), E, fft, outs, stream=stream)
        b.record()
        torch.cuda.synchronize()
        res.setdefault(n, []).append(a.elapsed_time(b) / 40)
for n, v in res.items():
    print(f"N={fft} tail groups={n * 256:5d}: " + " ".join(f"{x:.4f}" for x in v) + f"  frac={E*spe*8/(min(v)*1e-3)/8e12:.4f}")
