#!/usr/bin/env python3
"""What the fused fp64 ANN + cascade costs the reference-mode kernel: the same batch with decide =
ANN (the reference's configuration), THRESHOLD and NONE.  Interleaved repetitions on one box."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "cognitive-radio-network_amd"))
import torch
import crnsense as cs

dev = torch.device("cuda", 0)
base = cs.cfg_reference()
spe = cs.samples_per_epoch(base)
E = (28672 * 40960) // spe
iq = torch.zeros(cs.samples_needed(base, E) * 2, dtype=torch.float32, device=dev)
truth = torch.empty(E, dtype=torch.int32, device=dev)
feats = torch.empty(E, 4, dtype=torch.float32, device=dev)
occ = torch.empty(E, 4, dtype=torch.uint8, device=dev)
dec = torch.empty(E, dtype=torch.int32, device=dev)
ann = torch.empty(E, 3, dtype=torch.float64, device=dev)
stream = torch.cuda.current_stream().cuda_stream
outs = {"features": feats.data_ptr(), "ann_out": ann.data_ptr(), "decision": dec.data_ptr(), "occupancy": occ.data_ptr(), "spectrum": 0}
sensors = {}
for name, d in (("ANN", cs.DECIDE_ANN), ("THRESHOLD", cs.DECIDE_THRESHOLD), ("NONE", cs.DECIDE_NONE)):
    c = cs.cfg_reference()
    c.decide = d
    if d == cs.DECIDE_THRESHOLD:
        c.ref_band = 0
        for b in range(4):
            c.thresh[b] = 4.0
    sensors[name] = cs.Sensor(c)
sensors["ANN"].synth_fill_device(iq.data_ptr(), E, spe, seed=1, truth_ptr=truth.data_ptr(), stream=stream)
for _ in range(60):
    sensors["ANN"].run_device(iq.data_ptr(), E, 512, outs, stream=stream)
torch.cuda.synchronize()
res = {}
for rep in range(3):
    for name, s in sensors.items():
        for _ in range(5):
            s.run_device(iq.data_ptr(), E, 512, outs, stream=stream)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(40):
            s.run_device(iq.data_ptr(), E, 512, outs, stream=stream)
        b.record()
        torch.cuda.synchronize()
        res.setdefault(name, []).append(a.elapsed_time(b) / 40)
for name, v in res.items():
    print(f"ref512 decide={name:10s}: " + " ".join(f"{x:.4f}" for x in v) + f"  frac={E*spe*8/(min(v)*1e-3)/8e12:.4f}")
