#!/usr/bin/env python3
"""Where a 2 GiB launch loses time: epoch closes per 10 us of wall clock, stamped inside the kernel.

Variant 17 (A/B build) writes s_memrealtime (100 MHz, one clock for all XCDs) at every epoch close.  The number of closes per time
bin is the machine's throughput over the launch: a ramp, a plateau, and a tail in which the last, partly filled round of workgroups
runs.  What a launch could gain at most = 1 - epochs / (plateau rate x span): that is what a neighbouring launch on a second stream
fills (tools/gpu_streams_2gib.py measures it)."""
import os
import sys

os.environ.setdefault("CRN_SENSE_AB", "1")   # measurement variants: libcrnsense_ab.so
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "cognitive-radio-network_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import crnsense as cs  # noqa: E402

cfg = cs.cfg_energy_scaled(4096, 4.0)
spe = cs.samples_per_epoch(cfg)
dev = torch.device("cuda", 0)
s = cs.Sensor(cfg)
E_max = 28672
iq = torch.zeros(cs.samples_needed(cfg, E_max) * 2, dtype=torch.float32, device=dev)
stream = torch.cuda.current_stream().cuda_stream
s.synth_fill_device(iq.data_ptr(), E_max, spe, seed=1, stream=stream)
s.set_variant(17)
if os.environ.get("CRN_TIMELINE_GEOMETRY"):   # e.g. "208,104": geometry codes of crn_sense_set_variant, applied in order
    for code in os.environ["CRN_TIMELINE_GEOMETRY"].split(","):
        s.set_variant(int(code))
    print("geometry codes:", os.environ["CRN_TIMELINE_GEOMETRY"])
results = {}
for E in (28672, 13107, 6553):
    feats = torch.empty(E, cfg.n_bands, dtype=torch.float32, device=dev)
    occ = torch.empty(E, cfg.n_bands, dtype=torch.uint8, device=dev)
    dec = torch.empty(E, dtype=torch.int32, device=dev)
    tr = torch.zeros(E * 4, dtype=torch.int64, device=dev)   # [E][3] close stamps, then one start stamp per workgroup
    outs = {"features": feats.data_ptr(), "ann_out": tr.data_ptr(), "decision": dec.data_ptr(), "occupancy": occ.data_ptr(), "spectrum": 0}
    rows = []
    for rep in range(5):
        for _ in range(40):
            s.run_device(iq.data_ptr(), E, 4096, outs, stream=stream)
        torch.cuda.synchronize()
        tr[E * 3:].zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        s.run_device(iq.data_ptr(), E, 4096, outs, stream=stream)
        e1.record()
        torch.cuda.synchronize()
        a = tr.cpu().numpy()
        starts = a[E * 3:]
        starts = np.sort(starts[starts > 0])
        closes = np.sort(a[:E * 3].reshape(E, 3)[:, 0])
        rows.append((e0.elapsed_time(e1) * 1e3, (starts - starts[0]) / 100.0, (closes - starts[0]) / 100.0))
    results[E] = sorted(rows, key=lambda r: r[0])[len(rows) // 2]
# the steady rate: slope of the cumulative close count over the middle of the big batch (closes come in bursts: the workgroups of a
# round run in step, so a rate needs several rounds)
_, _, c_big = results[28672]
i0, i1 = int(0.15 * len(c_big)), int(0.80 * len(c_big))
rate = np.polyfit(c_big[i0:i1], np.arange(i0, i1), 1)[0]   # closes per us
print(f"steady rate in the middle of the 8.75 GiB launch (closes 15 % .. 80 %): {rate:.2f} epochs/us = {rate * 1e6 * spe * 8 / 8e12:.3f} of the HBM peak")
for E in (28672, 13107, 6553):
    us, st, cl = results[E]
    ideal = E / rate
    step = 1024 / rate   # one round of the 1024 workgroup slots at the steady rate
    hist = np.bincount((cl / step).astype(np.int64))
    n_first = min(1024, len(st))
    print(f"{E} epochs ({E * spe * 8 / 2**30:.2f} GiB), {len(st)} workgroups: {us:.1f} us by events = {E * spe * 8 / us * 1e6 / 8e12:.3f} of the peak; "
          f"at the steady rate {ideal:.1f} us: {100 * (1 - ideal / us):.1f} % of the launch is ramp and tail")
    print(f"   first workgroup start -> last close {cl[-1]:.1f} us (the events see {us - cl[-1]:.1f} us more: dispatch and completion); the first {n_first} workgroups start within "
          f"{st[n_first - 1]:.1f} us; first close at {cl[0]:.1f} us, the 1024th at {cl[min(1023, E - 1)]:.1f} us (a round at the steady rate: {step:.1f} us)")
    last = cl[-1] - cl[::-1]
    print(f"   drain: the last 1024 closes spread over {last[min(1023, E - 1)]:.1f} us, the last 512 over {last[511]:.1f} us, the last 128 over {last[127]:.1f} us "
          f"(at the steady rate: {1024 / rate:.1f} / {512 / rate:.1f} / {128 / rate:.1f} us)")
    print(f"   closes per {step:.1f} us from the first workgroup's start: " + " ".join(str(int(x)) for x in hist))
