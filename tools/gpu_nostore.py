#!/usr/bin/env python3
"""Where does the epoch close's remaining cost come from?  Same kernel, same batch, with (a) all
outputs, (b) no output pointers at all (the close computes everything, stores nothing), (c) the
no-close ablation (variant 16).  Interleaved repetitions on one box."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "cognitive-radio-network_amd"))
import torch
import os as _os
_os.environ.setdefault("CRN_SENSE_AB", "1")   # measurement variants: libcrnsense_ab.so
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
full = {"features": feats.data_ptr(), "ann_out": 0, "decision": dec.data_ptr(), "occupancy": occ.data_ptr(), "spectrum": 0}
none = {"features": 0, "ann_out": 0, "decision": 0, "occupancy": 0, "spectrum": 0}
only_occ = {"features": 0, "ann_out": 0, "decision": 0, "occupancy": occ.data_ptr(), "spectrum": 0}
for _ in range(60):
    s.run_device(iq.data_ptr(), E, fft, full, stream=stream)
torch.cuda.synchronize()
res = {}
cases = [("all outputs", 0, full), ("occupancy only", 0, only_occ), ("no outputs", 0, none)]
if fft == 4096:
    cases.append(("no close (v16)", 16, full))
    cases.append(("v16 + barrier", 18, full))
for rep in range(3):
    for name, var, outs in cases:
        s.set_variant(var)
        for _ in range(5):
            s.run_device(iq.data_ptr(), E, fft, outs, stream=stream)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(40):
            s.run_device(iq.data_ptr(), E, fft, outs, stream=stream)
        b.record()
        torch.cuda.synchronize()
        res.setdefault(name, []).append(a.elapsed_time(b) / 40)
for name, v in res.items():
    print(f"N={fft} {name:16s}: " + " ".join(f"{x:.4f}" for x in v) + f"  frac={E*spe*8/(min(v)*1e-3)/8e12:.4f}")
