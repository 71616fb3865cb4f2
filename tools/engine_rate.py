"""Decisions per second of the C++ engine driven packet by packet through tests/harness/engine_harness, and
the time spent inside execute() (CE_mutex held): default enqueue-only mode and the synchronous form (-a 0)."""
import os, sys, subprocess, time
sys.path[:0] = [os.path.join(os.getcwd(), "cognitive-radio-network_amd"), os.path.join(os.getcwd(), "tests")]
import numpy as np, crnsense as cs, signals
cfg = cs.cfg_reference()
L, n_epochs = 364, 4000
iq, picks = signals.make_epochs(cfg, n_epochs, seed=5, L=L)
iq.tofile("/tmp/iq_rate.bin")
for name, mode in (("enqueue-only (default)", []), ("synchronous (-a 0)", ["-a", "0"])):
    t0 = time.perf_counter()
    out = subprocess.run(["tests/harness/engine_harness", "/tmp/iq_rate.bin", str(L), "-g", "0", "-v", "0", "-s", "1"] + mode, capture_output=True, text=True, timeout=600)
    dt = time.perf_counter() - t0
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("epoch ")]
    dec = np.array([int(l.split()[3]) for l in lines])
    print(f"engine {name}: {len(lines)} epochs, {(dec == picks[:len(dec)]).mean():.4f} correct, {dt:.2f} s wall incl. process start -> {len(lines) / dt:.0f} decisions/s ({len(lines) * 10 * L / dt / 1e6:.1f} Msamples/s; the radio delivers 13 Msamples/s)")
    for ln in out.stdout.splitlines():
        if ln.startswith("execute_us") or ln.startswith("epoch_closing") or ln.startswith("CE_Predictive_Node_GPU:"):
            print("   ", ln)
