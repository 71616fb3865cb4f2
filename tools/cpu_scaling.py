"""How the CPU restatement scales with threads on this box (what bench.py's cpu_baseline `cores` means).
Test-side tool: the oracle is only timed here, never used by the product."""
import os, sys, time
sys.path[:0] = [os.path.join(os.getcwd(), "cognitive-radio-network_amd"), os.path.join(os.getcwd(), "tests")]
import numpy as np, crnsense as cs, oracle_py as orc, signals
print("os.cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
try:
    print("cpu.max", open("/sys/fs/cgroup/cpu.max").read().strip())
except Exception as e:
    print("cpu.max n/a", e)
print("native build:", orc.use_native_build())
cfg = cs.cfg_energy_scaled(4096, 4.0)
n = 2048
rng = np.random.default_rng(1)
iq = (rng.standard_normal(n * 40960 * 2) * 1e-3).astype(np.float32)
for th in (1, 2, 4, 8, 16, 32, 64, 128, 256):
    if th > (os.cpu_count() or 1):
        break
    orc.run(cfg, iq, min(n, 4 * th), n_threads=th)
    best = 0
    for _ in range(3):
        t = time.perf_counter(); orc.run(cfg, iq, n, n_threads=th); dt = time.perf_counter() - t
        best = max(best, n * 40960 / dt / 1e6)
    print(f"threads {th:4d}: {best:8.1f} Msamples/s  ({best / th:6.1f} per thread)")
