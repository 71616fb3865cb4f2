#!/usr/bin/env python3
"""The relayed end of a launch (SenseParams::n_relay_groups): outputs bit-identical to the unrelayed launch at every size, and what
it buys in kernel time.  600 + n = n x 64 relayed groups, 700 + S = S runs per group, 200 + n = n x 256 single-group tail workgroups."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "cognitive-radio-network_amd")]
import numpy as np  # noqa: E402
import torch  # noqa: E402

import crnsense as cs  # noqa: E402

dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream().cuda_stream


def outputs(cfg, E):
    nb = cfg.n_bands
    t = [torch.zeros(E, nb, device=dev), torch.zeros(E, 3, dtype=torch.float64, device=dev), torch.zeros(E, dtype=torch.int32, device=dev),
         torch.zeros(E, nb, dtype=torch.uint8, device=dev)]
    return t, {"features": t[0].data_ptr(), "ann_out": t[1].data_ptr(), "decision": t[2].data_ptr(), "occupancy": t[3].data_ptr(), "spectrum": 0}


def check():
    bad = 0
    for name, cfg in (("ref512", cs.cfg_reference()), ("e512", cs.cfg_energy_scaled(512, 4.0)), ("e1024", cs.cfg_energy_scaled(1024, 4.0)),
                      ("e2048", cs.cfg_energy_scaled(2048, 4.0)), ("e4096", cs.cfg_energy_scaled(4096, 4.0))):
        for K in (10, 7, 3):
            cfg.frames_per_epoch = K
            N = cfg.fft_len
            spe = cs.samples_per_epoch(cfg)
            for E in (1500 * 4096 // N + 3, 3001 * 4096 // N):
                s = cs.Sensor(cfg)
                L = N if name != "ref512" else 364
                # two different inputs in turn: a relay buffer left by the previous launch must never pass for this one's
                iqs, refs = [], []
                s.set_variant(600)
                for k in range(2):
                    iq = torch.zeros(cs.samples_needed(cfg, E) * 2, dtype=torch.float32, device=dev)
                    s.synth_fill_device(iq.data_ptr(), E, spe, seed=E + K + 1000 * k, stream=stream)
                    ref_t, ref_o = outputs(cfg, E)
                    s.run_device(iq.data_ptr(), E, L, ref_o, stream=stream)
                    iqs.append(iq)
                    refs.append(ref_t)
                torch.cuda.synchronize()
                assert not torch.equal(refs[0][0], refs[1][0])
                for groups, segs in ((1, 2), (4, 2), (8, 3), (16, 5), (16, 8), (3, 2), (32, 2)):
                    s.set_variant(600 + groups)
                    s.set_variant(700 + segs)
                    outs = [outputs(cfg, E) for _ in range(6)]
                    for rep in range(6):     # back to back, no host synchronisation in between
                        s.run_device(iqs[rep & 1].data_ptr(), E, L, outs[rep][1], stream=stream)
                    torch.cuda.synchronize()
                    for rep in range(6):
                        if not all(torch.equal(a, b) for a, b in zip(outs[rep][0], refs[rep & 1])):
                            bad += 1
                            print(f"MISMATCH {name} K={K} E={E} relay groups {groups * 64} runs {segs} launch {rep}")
                s.close()
        print(f"{name}: relayed launches identical to the plain ones" if bad == 0 else f"{name}: {bad} mismatches so far", flush=True)
    return bad


def timing(E, combos, reps=5, n=40):
    cfg = cs.cfg_energy_scaled(4096, 4.0)
    spe = cs.samples_per_epoch(cfg)
    s = cs.Sensor(cfg)
    iq = torch.zeros(cs.samples_needed(cfg, E) * 2, dtype=torch.float32, device=dev)
    s.synth_fill_device(iq.data_ptr(), E, spe, seed=7, stream=stream)
    t, o = outputs(cfg, E)
    res = {c: [] for c in combos}
    for rep in range(reps):
        for c in combos:
            tail, groups, segs = c
            s.set_variant(200 + tail)
            s.set_variant(600 + groups)
            s.set_variant(700 + segs)
            for _ in range(15):
                s.run_device(iq.data_ptr(), E, 4096, o, stream=stream)
            evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
            for a, b in evs:
                a.record()
                s.run_device(iq.data_ptr(), E, 4096, o, stream=stream)
                b.record()
            torch.cuda.synchronize()
            res[c].append(float(np.mean([a.elapsed_time(b) for a, b in evs])))
    print(f"--- {E} epochs ({E * spe * 8 / 2**30:.2f} GiB), 4096-pt headline kernel; mean of {n} launches, {reps} interleaved repetitions")
    for c in combos:
        m = float(np.median(res[c]))
        print(f"tail {c[0] * 256:5d} singles, relay {c[1] * 64:5d} groups x {c[2]} runs: {m * 1e3:8.1f} us = {E * spe * 8 / (m * 1e-3) / 8e12:.4f} of the HBM peak   "
              + " ".join(f"{x * 1e3:.1f}" for x in res[c]))
    s.close()


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    if what in ("all", "check"):
        if check():
            sys.exit(1)
    if what in ("all", "time"):
        combos = [(4, 0, 2), (4, 4, 2), (4, 8, 2), (4, 12, 2), (4, 16, 2), (2, 8, 2), (2, 16, 2), (0, 16, 2), (4, 8, 3), (4, 16, 3), (8, 8, 2), (8, 0, 2)]
        timing(28672, combos)
        timing(6553, combos)
        timing(13107, combos)
