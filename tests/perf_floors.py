"""Throughput floors of the hot path — north_star's ">= 70 % of the HBM-read roofline" as a TESTED property.

The kernels are frozen; these numbers are what "frozen" means for speed.  Each floor sits 8 % under the LOWEST value
any driver run (BENCH_r01..r05) or committed profile has shown for that leg, on the driver's own shape (`bench.py --steps 20 --warmup 5`,
default batch) — the lowest being, for every leg, the slowest box of the round-6 suite's five runs
(profiles/r06_roofline_floors_fourth_box.json: 2 - 4 % under the others, the same binary) — so that box-to-box and run-to-run spread
(power cap, HBM temperature) never trips them and a real regression — a spilled register, a lost workgroup per CU, a re-read of the input, a launch geometry that
leaves slots empty — always does.  tests/test_zz_roofline_floors.py asserts them on the GPU; profiles/r06_floor_gate_can_fail.txt shows the
same assertions going red on a deliberately bad launch geometry.

The hot loop they protect: /root/reference/cognitive_engines/CE_Predictive_Node/CE_Predictive_Node.cpp:150-154 (fft_execute + the
magnitude accumulation), batched.

    leg                                   lowest seen (where)                                   floor
"""
FLOORS = {
    # bench.py (headline: 4096-pt x 3ch, 8.75 GiB batch, pruned to the reference channel plan)
    "headline": 0.75,               # 0.8136 (round 6, fourth box; BENCH_r01: 0.814); 0.821 .. 0.849 elsewhere
    "alt.cfgH_2GiB_batch": 0.71,    # 0.7755 (round 6, fourth box; BENCH_r05: 0.787): SURVEY.md §8(d) cfgH's batch as worded, one stream
    "alt.cfgH_2GiB_batch_two_streams": 0.73,   # 0.7931 (round 6, fourth box; 0.808 .. 0.828 elsewhere)
    "alt.unpruned": 0.72,           # 0.783 (round 6, fourth box; BENCH_r05: 0.794): what any other band plan or a spectrum request runs
    # the other BASELINE.json configurations, same shape (--no-alt)
    "ref512": 0.755,                # 0.821 (round 6, fourth box; 0.831 .. 0.837 elsewhere): cfg3, reference-exact + ANN
    "energy1024": 0.77,             # 0.8377 (round 6, fourth box; 0.848 .. 0.851 elsewhere): cfg1
    "welch4096": 0.435,             # 0.4751 (round 6, fourth box; 0.490 .. 0.495 elsewhere): cfg2 — VALU-issue-bound, not HBM-bound
}
NORTH_STAR = 0.70                   # BASELINE.json north_star: ">= 70 % HBM-read roofline on 4096-pt batched FFT+energy at 1 GPU"
DRIVER_SHAPE = ("--steps", "20", "--warmup", "5")
ATTEMPTS = 3                        # a leg under its floor is re-measured in a fresh process: best of 3 (a 2x regression fails all three)
