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
        if ln.startswith(("execute_us", "control_two_clock_reads_us", "epoch_closing", "CE_Predictive_Node_GPU:")):
            print("   ", ln)
# the extension modes at the engine's default size (512 points): Welch estimate on the reference's channels, and the 64-band scan with
# its start-up calibration — one epoch per launch as well (windowed, overlapped frames: the dealt-frame kernel's windowed forms)
rng = np.random.default_rng(6)
for name, mode, K in (("-m welch", ["-m", "welch"], 10), ("-m scan -c 8", ["-m", "scan", "-c", "8"], 8)):
    P = -(-((K - 1) * 256 + 512) // L)
    n_ep = 2000
    noise = rng.normal(0, 7e-4, n_ep * P * L * 2).astype(np.float32)
    noise.tofile("/tmp/iq_rate_w.bin")
    out = subprocess.run(["tests/harness/engine_harness", "/tmp/iq_rate_w.bin", str(L), "-g", "0", "-v", "0", "-s", "1"] + mode, capture_output=True, text=True, timeout=600)
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("epoch ")]
    print(f"engine {name} (enqueue-only, {P} packets of {L} samples per epoch): {len(lines)} epochs decided")
    for ln in out.stdout.splitlines():
        if ln.startswith("epoch_closing") or ln.startswith("CE_Predictive_Node_GPU:"):
            print("   ", ln)
