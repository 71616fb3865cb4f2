"""Throughput floors of the hot path — north_star's ">= 70 % of the HBM-read roofline" as a TESTED property.

The kernels are frozen; these numbers are what "frozen" means for speed.  Each floor sits about 8 % under the LOWEST value
any driver run (BENCH_r01..r05) or committed profile has shown for that leg, on the driver's own shape (`bench.py --steps 20 --warmup 5`,
default batch), so that box-to-box and run-to-run spread (power cap, HBM temperature: +-1.5 %, profiles/r05_bench_driver_shape*.json)
never trips them and a real regression — a spilled register, a lost workgroup per CU, a re-read of the input, a launch geometry that
leaves slots empty — always does.  tests/test_zz_roofline_floors.py asserts them on the GPU; profiles/r06_floor_gate_can_fail.txt shows the
same assertions going red on a deliberately bad launch geometry.

The hot loop they protect: /root/reference/cognitive_engines/CE_Predictive_Node/CE_Predictive_Node.cpp:150-154 (fft_execute + the
magnitude accumulation), batched.

    leg                                   lowest seen (where)                                   floor
"""
FLOORS = {
    # bench.py (headline: 4096-pt x 3ch, 8.75 GiB batch, pruned to the reference channel plan)
    "headline": 0.76,               # 0.814 (BENCH_r01); 0.821 / 0.825 / 0.822 / 0.827 after
    "alt.cfgH_2GiB_batch": 0.72,    # 0.787 (BENCH_r05): SURVEY.md §8(d) cfgH's batch as worded, one stream
    "alt.cfgH_2GiB_batch_two_streams": 0.75,   # 0.808 (profiles/r05_bench_driver_shape.json)
    "alt.unpruned": 0.73,           # 0.794 (BENCH_r05): what any other band plan or a spectrum request runs
    # the other BASELINE.json configurations, same shape (--no-alt)
    "ref512": 0.76,                 # 0.831 (profiles/r05_bench_cfg3_ref512.json): cfg3, reference-exact + ANN
    "energy1024": 0.78,             # 0.851 (profiles/r05_bench_cfg1_1024pt.json): cfg1
    "welch4096": 0.45,              # 0.491 (profiles/r05_bench_cfg2_welch.json): cfg2 — VALU-issue-bound, not HBM-bound
}
NORTH_STAR = 0.70                   # BASELINE.json north_star: ">= 70 % HBM-read roofline on 4096-pt batched FFT+energy at 1 GPU"
DRIVER_SHAPE = ("--steps", "20", "--warmup", "5")
ATTEMPTS = 3                        # a leg under its floor is re-measured in a fresh process: best of 3 (a 2x regression fails all three)
