#!/usr/bin/env python3
"""Run-time row pruning (variant 24, kRowsRT) against the compile-time pruned default (13) and the full kernel (2): headline workload
with the reference plan and with another sparse plan; outputs compared bit for bit.  Interleaved repetitions on one box."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "cognitive-radio-network_amd")]
os.environ.setdefault("CRN_SENSE_AB", "1")   # variant 24 is a measurement variant: libcrnsense_ab.so
import crnsense as cs  # noqa: E402

dev = torch.device("cuda", 0)
E = 28672


def plan(kind):
    cfg = cs.cfg_energy_scaled(4096, 4.0)
    if kind == "other":     # four bands in rows {3, 4, 10, 11, 12}: outside the reference plan's rows
        segs = [(800, 1000, 0), (1030, 1200, 1), (2600, 2900, 2), (2950, 3200, 3)]
        for i, (lo, hi, b) in enumerate(segs):
            cfg.segs[i].lo, cfg.segs[i].hi, cfg.segs[i].band = lo, hi, b
        cfg.n_segs = 4
    if kind == "dense":     # 16 bands, one per row: nothing to prune
        for b in range(16):
            cfg.segs[b].lo, cfg.segs[b].hi, cfg.segs[b].band = 256 * b + 10, 256 * b + 200, b
        cfg.n_segs, cfg.n_bands, cfg.ref_band = 16, 16, -1
        for b in range(16):
            cfg.thresh[b] = 1.0
    return cfg


spe = 40960
iq = torch.zeros(E * spe * 2, dtype=torch.float32, device=dev)
s0 = cs.Sensor(plan("ref"))
s0.synth_fill_device(iq.data_ptr(), E, spe, seed=0xC0FFEE)
stream = torch.cuda.current_stream().cuda_stream
for kind in ("ref", "other", "dense"):
    cfg = plan(kind)
    sensors, outs = {}, {}
    for v in (13, 2, 24):
        s = cs.Sensor(cfg)
        s.set_variant(v)
        sensors[v] = s
        f = torch.empty(E, cfg.n_bands, device=dev)
        o = torch.empty(E, cfg.n_bands, dtype=torch.uint8, device=dev)
        d = torch.empty(E, dtype=torch.int32, device=dev)
        outs[v] = (f, o, d, {"features": f.data_ptr(), "ann_out": 0, "decision": d.data_ptr(), "occupancy": o.data_ptr(), "spectrum": 0})
    ms = {v: [] for v in sensors}
    for rep in range(4):
        for v, s in sensors.items():
            for _ in range(8):
                s.run_device(iq.data_ptr(), E, 4096, outs[v][3], stream=stream)
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(15)]
            for a, b in ev:
                a.record()
                s.run_device(iq.data_ptr(), E, 4096, outs[v][3], stream=stream)
                b.record()
            torch.cuda.synchronize()
            ms[v] += [a.elapsed_time(b) for a, b in ev]
    same = all(torch.equal(outs[2][i], outs[v][i]) for v in (13, 24) for i in range(3))
    print(f"plan '{kind}': " + "  ".join(f"v{v} {np.mean(ms[v]):.4f} ms = {E * spe * 8 / (np.mean(ms[v]) * 1e-3) / 8e12:.4f} ({sensors[v].kernel_info()['name'][-40:]})"
                                          for v in sensors) + f"   outputs identical: {same}")
