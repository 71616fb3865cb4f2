#!/usr/bin/env python3
"""The engine the way a CRTS node runs it: tests/harness/ecr_threads plays the ECR's rx worker and CE worker as two threads with the
reference's locking, packets paced at the radio's rate (364 samples / 13 Msps = 28 us), one sensing epoch per 100 ms, GPU and
launcher thread idle in between.  Prints the engine's own counters (-s 1): kernel time and the time from the epoch's last packet to
a readable decision — with the ring's pre-wake (the launcher thread is told ten packets ahead of the hand-off) and without it — and
the distribution of execute()'s duration: its ten longest calls attributed (event, packet of the epoch, decision reported or not)
next to a control (two clock reads with nothing between them: what the operating system does to the thread), over five runs.
(profiles/r05_engine_idle_gap_vs_r04_library.txt: the same with three runs of the round-4 library preloaded, taken before the ABI
version moved to 4.)

    python3 tools/engine_idle_gap.py
"""
import os, subprocess, sys
sys.path[:0] = [os.path.join(os.getcwd(), "cognitive-radio-network_amd"), os.path.join(os.getcwd(), "tests")]
import numpy as np, crnsense as cs, signals
cfg = cs.cfg_reference()
L, per_seg = 364, 64
segs = []
for ch in range(4):
    iq, _ = signals.make_epochs(cfg, 7, seed=900 + ch, L=L, picks=[ch] * 7)
    segs.append(iq[: per_seg * L * 2])
np.concatenate(segs).tofile("/tmp/iq_gap.bin")
runs = [("this build, pre-wake on (default)", {})] * 5 + [("this build, pre-wake off (CRN_INGEST_PREWAKE_US=0)", {"CRN_INGEST_PREWAKE_US": "0"})] * 2
runs += [("this build, pre-wake without the empty warm-up launch (CRN_INGEST_WARM_GPU=0)", {"CRN_INGEST_WARM_GPU": "0"})]
worst = {}
for name, env in runs:
    out = subprocess.run(["tests/harness/ecr_threads", "/tmp/iq_gap.bin", str(L), str(per_seg), "4.1", "-v", "0", "-s", "1"],
                         capture_output=True, text=True, timeout=120, env=dict(os.environ, CRN_INGEST_TRACE="1", **env))
    n = len([ln for ln in out.stdout.splitlines() if ln.startswith("decision ")])
    print(f"engine between the ECR's threads, packets every 28 us, one epoch per 100 ms, {name}: {n} decisions")
    for ln in out.stdout.splitlines():
        if ln.startswith(("CE_Predictive_Node_GPU:", "execute_us", "control_two_clock_reads_us", "longest_execute")):
            print("   ", ln)
        if ln.startswith("execute_us"):
            w = worst.setdefault(name, [0, 0.0, 0.0])
            w[0] += int(ln.split()[2])
            w[1] = max(w[1], float(ln.split()[-1]))
        if ln.startswith("control_two_clock_reads_us"):
            worst[name][2] = max(worst[name][2], float(ln.split()[-1]))
    for ln in out.stderr.splitlines():
        if ln.startswith("crn_ingest trace"):
            print("   ", ln)
print("summary: calls, longest execute(), longest gap between two clock reads with nothing between them (the OS alone)")
for name, (calls, mx, ctl) in worst.items():
    print(f"    {name}: {calls} calls, max execute() {mx:.1f} us, max control {ctl:.1f} us")
