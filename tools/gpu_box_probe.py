#!/usr/bin/env python3
"""Which kernel variant does this box like?  The pool has boxes whose per-launch times swing widely
under the sensing kernel.  Prints the headline kernel's time statistics and a few
variants interleaved, plus a pure streaming read, so that a run which lands on such a box shows
whether a lighter variant (3 workgroups per CU, no row pruning, ...) holds up better there."""
import os, sys, statistics as st
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "cognitive-radio-network_amd"))
import torch
import os as _os
_os.environ.setdefault("CRN_SENSE_AB", "1")   # measurement variants: libcrnsense_ab.so
import crnsense as cs

fft = 4096
cfg = cs.cfg_energy_scaled(fft, 4.0)
spe = cs.samples_per_epoch(cfg)
E = 28672
dev = torch.device("cuda", 0)
s = cs.Sensor(cfg)
iq = torch.zeros(cs.samples_needed(cfg, E) * 2, dtype=torch.float32, device=dev)
feats = torch.empty(E, cfg.n_bands, dtype=torch.float32, device=dev)
occ = torch.empty(E, cfg.n_bands, dtype=torch.uint8, device=dev)
dec = torch.empty(E, dtype=torch.int32, device=dev)
stream = torch.cuda.current_stream().cuda_stream
s.synth_fill_device(iq.data_ptr(), E, spe, seed=1, stream=stream)
outs = {"features": feats.data_ptr(), "ann_out": 0, "decision": dec.data_ptr(), "occupancy": occ.data_ptr(), "spectrum": 0}
for _ in range(60):
    s.run_device(iq.data_ptr(), E, fft, outs, stream=stream)
torch.cuda.synchronize()
# per-launch spread of the default kernel
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(100)]
for a, b in ev:
    a.record(); s.run_device(iq.data_ptr(), E, fft, outs, stream=stream); b.record()
torch.cuda.synchronize()
t = [a.elapsed_time(b) for a, b in ev]
print(f"default kernel per launch: mean {st.mean(t):.4f} ms  min {min(t):.4f}  max {max(t):.4f}  sd {st.pstdev(t):.4f}  -> {E*spe*8/(st.mean(t)*1e-3)/8e12:.4f}")
res = {}
names = {13: "default (4 WG/CU, pruned)", 2: "4 WG/CU, unpruned", 7: "default w/o priority", 10: "3 WG/CU, all twiddles in regs", 1: "4 WG/CU, no prefetch", 16: "no epoch close (ablation)"}
for rep in range(3):
    for v in names:
        s.set_variant(v)
        for _ in range(5):
            s.run_device(iq.data_ptr(), E, fft, outs, stream=stream)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(40):
            s.run_device(iq.data_ptr(), E, fft, outs, stream=stream)
        b.record()
        torch.cuda.synchronize()
        res.setdefault(v, []).append(a.elapsed_time(b) / 40)
for v, x in res.items():
    print(f"variant {v:2d} {names[v]:32s}: " + " ".join(f"{y:.4f}" for y in x) + f"  frac={E*spe*8/(st.mean(x)*1e-3)/8e12:.4f}")
