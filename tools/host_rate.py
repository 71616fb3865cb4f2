#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer entry point (crn_sense_run_host: H2D + kernel + D2H +
sync), for the note in DESIGN.md — never the bench `value`."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "cognitive-radio-network_amd"), os.path.join(ROOT, "tests")]
import crnsense as cs  # noqa: E402

cfg = cs.cfg_energy_scaled(4096, 4.0)
E = 1024
iq = (np.random.default_rng(0).normal(0, 1e-3, E * 10 * 4096 * 2)).astype(np.float32)
s = cs.Sensor(cfg)
s.run_host(iq, E)
t = time.perf_counter()
reps = 5
for _ in range(reps):
    s.run_host(iq, E)
dt = (time.perf_counter() - t) / reps
print(f"run_host {E} epochs ({iq.nbytes / 2**20:.0f} MiB pageable host memory): {dt * 1e3:.1f} ms per call = "
      f"{E * 40960 / dt / 1e6:.0f} Msamples/s = {iq.nbytes / dt / 1e9:.1f} GB/s incl. PCIe")
# one reference-sized epoch, as the engine issues it (10 packets of 364 samples)
ref = cs.Sensor(cs.cfg_reference())
one = iq[: 10 * 364 * 2]
ref.run_host(one, 1, L=364)
t = time.perf_counter()
for _ in range(200):
    ref.run_host(one, 1, L=364)
print(f"run_host 1 reference epoch (10 x 364 samples): {(time.perf_counter() - t) / 200 * 1e6:.0f} us per decision")
