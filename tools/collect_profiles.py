#!/usr/bin/env python3
"""Copy the round's evidence from gpurun_out/<tag> (scratch) into profiles/ (tracked) and derive
profiles/hbm_traffic.json from the FETCH_SIZE / WRITE_SIZE passes."""
import collections, csv, glob, json, os, shutil, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
SRC = os.path.join(R, "gpurun_out", tag)
DST = os.path.join(R, "profiles")
os.makedirs(DST, exist_ok=True)
for src, dst in (("stats/runc/*_kernel_stats.csv", f"{tag}_kernel_stats.csv"), ("bench_headline.json", f"{tag}_bench_headline.json"),
                 ("bench_cfg1_1024.json", f"{tag}_bench_cfg1_1024pt.json"), ("bench_cfg3_ref512.json", f"{tag}_bench_cfg3_ref512.json"),
                 ("bench_cfg2_welch.json", f"{tag}_bench_cfg2_welch.json"), ("host_rate.txt", f"{tag}_host_buffer_rate.txt"),
                 ("bench_unpruned.json", f"{tag}_bench_headline_unpruned.json"),
                 ("bench_e512.json", f"{tag}_bench_energy_512pt.json"), ("bench_e2048.json", f"{tag}_bench_energy_2048pt.json"),
                 ("membw_policy.txt", f"{tag}_stream_ceiling_policy.txt"), ("cfg3_decisions.txt", f"{tag}_cfg3_decisions_1M_epochs.txt"),
                 ("pytest_gpu.log", f"{tag}_pytest_gpu.log"), ("smoke.log", f"{tag}_smoke.log"),
                 ("bench_cfg2_welch_K32.json", f"{tag}_bench_cfg2_welch_K32.json"), ("bench_cfg4_scan_1rank.json", f"{tag}_bench_cfg4_scan_one_rank_rccl.json"),
                 ("engine_rate.txt", f"{tag}_engine_rate.txt"), ("zeros_probe.txt", f"{tag}_zero_input_probe.txt"),
                 ("power_probe.txt", f"{tag}_power_clock_probe.txt"), ("cpu_scaling.txt", f"{tag}_cpu_thread_scaling.txt"),
                 ("per_bin_error_at_floor.txt", f"{tag}_per_bin_error_at_floor.txt"),
                 ("engine_execute_latency.txt", f"{tag}_engine_execute_latency.txt"),
                 ("ring_rate.txt", f"{tag}_ring_rate_round.txt"),
                 ("bench_strong_share_per_gpu.jsonl", f"{tag}_bench_strong_share_per_gpu.jsonl"),
                 ("bench_strong_share_per_gpu_one_stream.jsonl", f"{tag}_bench_strong_share_per_gpu_one_stream.jsonl"),
                 ("bench_8ranks_one_gpu_strong.json", f"{tag}_bench_eight_self_launched_ranks_one_gpu_stand_in_wire_strong.json"),
                 ("decision_band.txt", f"{tag}_decision_band.txt"),
                 ("bench_driver_shape.json", f"{tag}_bench_driver_shape.json"),
                 ("bench_8ranks_one_gpu.json", f"{tag}_bench_eight_self_launched_ranks_one_gpu_stand_in_wire.json"),
                 ("bench_8ranks_one_gpu_scan.json", f"{tag}_bench_cfg4_eight_self_launched_ranks_one_gpu_stand_in_wire.json"),
                 ("welch_spans.txt", f"{tag}_welch_spans.txt"), ("soak.txt", f"{tag}_soak_round.txt"),
                 ("dealt_frames_ab.txt", f"{tag}_dealt_frames_ab.txt"), ("engine_idle_gap.txt", f"{tag}_engine_idle_gap.txt"),
                 ("per_bin_error_vs_snr.txt", f"{tag}_per_bin_error_vs_snr.txt"),
                 ("engine_between_ecr_threads.txt", f"{tag}_engine_between_ecr_threads.txt"),
                 ("asm_filter_byte_equal.txt", f"{tag}_asm_filter_byte_equal.txt"),
                 ("bench_wire_format_optional_library.json", f"{tag}_bench_wire_format_optional_library.json"),
                 ("bench_2ranks_one_gpu.json", f"{tag}_bench_two_ranks_one_gpu_stand_in_wire.json"),
                 ("two_real_rccl_ranks_one_gpu.txt", f"{tag}_two_real_rccl_ranks_one_gpu.txt"),
                 ("hipgraph_small_launches.txt", f"{tag}_hipgraph_small_launches.txt")):
    m = sorted(glob.glob(os.path.join(SRC, src)), key=os.path.getmtime)
    if m:
        shutil.copy(m[-1], os.path.join(DST, dst))
open(os.path.join(DST, f"{tag}_pmc_hbm_traffic.txt"), "w").write(
    subprocess.run([sys.executable, os.path.join(R, "tools", "pmc_summary.py"), SRC], capture_output=True, text=True).stdout)


def mean_counter(sub, name):
    f = sorted(glob.glob(os.path.join(SRC, sub, "*", "*_counter_collection.csv")), key=os.path.getmtime)
    if not f:
        return None
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f[-1]))
         if "sense_kernel" in r["Kernel_Name"] and r["Counter_Name"] == name]
    return sum(v[-5:]) / len(v[-5:]) if v else None


HOW = ("rocprofv3 --pmc FETCH_SIZE [GRBM_GUI_ACTIVE] and --pmc WRITE_SIZE [TCC_HIT_sum TCC_MISS_sum] in separate passes "
       "(--kernel-trace only) over `python3 bench.py --steps 5 --warmup 20 --cpu-epochs 0 <workload flags>`; FETCH_SIZE is KiB and on "
       "gfx950 tallies each 128-B request as 64 B, so x2 (MI355X_MICROARCH.md, HBM section); the x2 was re-calibrated for "
       "this kernel's 8-B-per-lane loads in round 1 (docs/history: the load instruction is unchanged since)")
out = {"_how": HOW}
# key = bench.py's f"{mode}{N}"; the unpruned 4096-pt kernel is recorded for the record only
for key, sub, bench in (("energy4096", "_headline", "bench_headline.json"), ("energy1024", "_cfg1", "bench_cfg1_1024.json"),
                        ("ref512", "_cfg3", "bench_cfg3_ref512.json"), ("welch4096", "_cfg2", "bench_cfg2_welch.json"),
                        ("energy512", "_e512", "bench_e512.json"), ("energy2048", "_e2048", "bench_e2048.json"),
                        ("energy4096_unpruned", "_unpruned", "bench_unpruned.json"), ("welch4096_K32", "_cfg2K32", "bench_cfg2_welch_K32.json")):
    fetch, write = mean_counter("pmc_fetch" + sub, "FETCH_SIZE"), mean_counter("pmc_write" + sub, "WRITE_SIZE")
    bj = os.path.join(SRC, bench)
    if fetch is None or not os.path.exists(bj):
        continue
    try:
        head = json.load(open(bj))
    except Exception:
        continue
    algo = head["config"]["bytes_per_gpu_per_step"]
    out[key] = {"epochs": head["config"]["epochs_per_gpu"], "FETCH_SIZE_KiB": fetch, "WRITE_SIZE_KiB": write,
                "hbm_bytes_per_launch": int(fetch * 1024 * 2 + (write or 0) * 1024),
                "algorithmic_bytes_per_launch": algo,
                # reads against the 8 B per unique input sample; everything (the per-epoch outputs are written once: features, occupancy,
                # decision, network outputs — 0.01 % of the bytes for the 4-band plans, 0.25 % for the 64-band scan) against the same
                "read_over_algorithmic": fetch * 1024 * 2 / algo,
                "total_over_algorithmic": (fetch * 1024 * 2 + (write or 0) * 1024) / algo}
    print(key, out[key])
if len(out) > 1:
    json.dump(out, open(os.path.join(DST, "hbm_traffic.json"), "w"), indent=1)
subprocess.run([sys.executable, os.path.join(R, "tools", "collect_valu_counters.py"), tag,
                f"energy4096=pmc_{tag}_headline:", f"welch4096=pmc_{tag}_welch:--mode welch", f"energy2048=pmc_{tag}_e2048:--fft 2048",
                f"ref512=pmc_{tag}_ref:--mode ref", f"energy1024=pmc_{tag}_e1024:--fft 1024", f"energy4096unpruned=pmc_{tag}_unpruned:--variant 2"])
for name in ("headline", "welch", "e2048", "ref", "e1024", "unpruned"):
    src = os.path.join(SRC, f"pmc_sq_{name}.txt")
    if os.path.exists(src):
        shutil.copy(src, os.path.join(DST, f"{tag}_pmc_sq_{name}.txt"))
print(sorted(os.listdir(DST)))
