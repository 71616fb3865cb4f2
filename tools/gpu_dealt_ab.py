#!/usr/bin/env python3
"""Small launches: the dealt-frame form of the sensing kernel (one epoch per workgroup, frames spread over its lane groups) against
the streaming form (an epoch per lane group, frames one after the other), per launch size.  Device time per launch from HIP events
around 200 back-to-back launches on one stream (input resident in HBM), and for the engine's shape also with the kernel reading
pinned host memory (what the ingest ring's one-epoch launch does).  Prints a table; with $CRN_EVIDENCE_DIR writes dealt_frames_ab.txt.

    python3 tools/gpu_dealt_ab.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cognitive-radio-network_amd"))
import crnsense as cs  # noqa: E402

dev = torch.device("cuda", 0)
REPS = 200


def time_launches(s, ptr, E, L, outs):
    for _ in range(20):
        s.run_device(ptr, E, L, outs)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(REPS):
        s.run_device(ptr, E, L, outs)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / REPS * 1e3   # us per launch


def single_launch_us(s, ptr, E, L, outs):
    """One launch at a time (the engine's pattern: nothing queued behind it): events around a lone launch, median of 100."""
    t = []
    for _ in range(110):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        s.run_device(ptr, E, L, outs)
        b.record()
        torch.cuda.synchronize()
        t.append(a.elapsed_time(b) * 1e3)
    return float(np.median(t[10:]))


lines = []
for label, cfg, L in (("reference 512-pt, |X| + network, 364-sample packets (the engine's shape)", cs.cfg_reference(), 364),
                      ("512-pt Welch scan: periodic Hann, hop 256, K = 8, 64 bands (the engine's -m scan at its default size)", cs.cfg_welch(512, 8, 64), 512),
                      ("512-pt energy detect, whole frames", cs.cfg_energy_scaled(512, 4.0), 512),
                      ("1024-pt energy detect, whole frames", cs.cfg_energy_scaled(1024, 4.0), 1024)):
    lines.append(label)
    lines.append(f"  {'epochs':>7} {'streaming us':>13} {'dealt us':>9} {'ratio':>6}   {'lone launch: streaming us':>26} {'dealt us':>9}")
    for b in range(cfg.n_bands):
        if cfg.decide == cs.DECIDE_THRESHOLD and cfg.ref_band < 0:
            cfg.thresh[b] = 1e-3
    for E in (1, 2, 8, 32, 128, 256, 512, 1024, 2048, 4096):
        need = cs.samples_needed(cfg, E, L)
        iq = (torch.randn(need * 2, device=dev) * 1e-2).contiguous()
        feats = torch.zeros(E, cfg.n_bands, device=dev)
        ann = torch.zeros(E, 3, dtype=torch.float64, device=dev)
        dec = torch.zeros(E, dtype=torch.int32, device=dev)
        occ = torch.zeros(E, cfg.n_bands, dtype=torch.uint8, device=dev)
        outs = {"features": feats.data_ptr(), "ann_out": ann.data_ptr(), "decision": dec.data_ptr(), "occupancy": occ.data_ptr(), "spectrum": 0}
        r = {}
        for code in (401, 402):
            s = cs.Sensor(cfg)
            s.set_variant(code)
            r[code] = (time_launches(s, iq.data_ptr(), E, L, outs), single_launch_us(s, iq.data_ptr(), E, L, outs))
            s.close()
        lines.append(f"  {E:>7} {r[401][0]:>13.2f} {r[402][0]:>9.2f} {r[401][0] / r[402][0]:>6.2f}   {r[401][1]:>26.2f} {r[402][1]:>9.2f}")
    if L == 364:
        # the ring's one-epoch launch: samples and results in pinned host memory, read / written by the kernel over the bus
        E = 1
        need = cs.samples_needed(cfg, E, L)
        h_iq = (torch.randn(need * 2) * 1e-2).pin_memory()
        h_feats = torch.zeros(E, cfg.n_bands).pin_memory()
        h_ann = torch.zeros(E, 3, dtype=torch.float64).pin_memory()
        h_dec = torch.zeros(E, dtype=torch.int32).pin_memory()
        h_occ = torch.zeros(E, cfg.n_bands, dtype=torch.uint8).pin_memory()
        outs = {"features": h_feats.data_ptr(), "ann_out": h_ann.data_ptr(), "decision": h_dec.data_ptr(), "occupancy": h_occ.data_ptr(), "spectrum": 0}
        r = {}
        for code in (401, 402):
            s = cs.Sensor(cfg)
            s.set_variant(code)
            r[code] = single_launch_us(s, h_iq.data_ptr(), E, L, outs)
            s.close()
        lines.append(f"  one epoch, samples and results in pinned host memory (the ring's launch): streaming {r[401]:.2f} us, dealt {r[402]:.2f} us")
text = "\n".join(lines)
print(text)
out_dir = os.environ.get("CRN_EVIDENCE_DIR")
if out_dir and os.path.isdir(out_dir):
    open(os.path.join(out_dir, "dealt_frames_ab.txt"), "w").write(text + "\n")
