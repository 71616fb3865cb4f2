#!/usr/bin/env python3
"""profiles/valu_counters.json from the SQ / GRBM counter passes of tools/gpu_pmc.sh (gpurun_out/pmc_<name>):
per workload the measured clock (GRBM_GUI_ACTIVE / 8 XCDs / kernel time), the VALU-busy fraction at that clock
(SQ_ACTIVE_INST_VALU is in quad-cycles: x 4 / (1024 SIMDs x kernel cycles)), VALU instructions per wave per frame,
and the LDS-wait share of wave cycles.  bench.py prints these beside the VALU roofline of the windowed modes.

  python tools/collect_valu_counters.py r02 welch4096=pmc_welch:--mode\ welch energy4096=pmc_v0: ..."""
import collections, csv, glob, json, os, sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
out = {"_how": "rocprofv3 --pmc passes (one counter group per run, --kernel-trace only) over `python3 bench.py --steps 5 --warmup 40 "
               "--cpu-epochs 0 <flags>` (tools/gpu_pmc.sh); means over the timed launches of sense_kernel. clock_ghz = GRBM_GUI_ACTIVE / 8 / "
               "kernel time (MI355X_MICROARCH.md, DVFS give-back); valu_busy_frac = SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs x kernel cycles)"}


def counters(d):
    acc = collections.defaultdict(list)
    dur = []
    for g in sorted(glob.glob(os.path.join(d, "*/"))):
        cc = sorted(glob.glob(g + "*/*_counter_collection.csv"), key=os.path.getmtime)[-1:]
        kt = sorted(glob.glob(g + "*/*_kernel_trace.csv"), key=os.path.getmtime)[-1:]
        if not cc:
            continue
        per = collections.defaultdict(list)
        for r in csv.DictReader(open(cc[0])):
            if "sense_kernel" in r["Kernel_Name"]:
                per[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in per.items():
            acc[k] = sum(v[-5:]) / len(v[-5:])
        if kt and os.path.basename(g.rstrip("/")) == "c":
            dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e9 for r in csv.DictReader(open(kt[0]))
                   if "sense_kernel" in r["Kernel_Name"]][-5:]
    return acc, (sum(dur) / len(dur) if dur else None)


for spec in sys.argv[2:]:
    key, rest = spec.split("=", 1)
    sub, flags = rest.split(":", 1)
    c, t = counters(os.path.join(R, "gpurun_out", sub))
    if not c or t is None:
        print("skip", key)
        continue
    n = int("".join(ch for ch in key if ch.isdigit()))
    mode = "".join(ch for ch in key if not ch.isdigit())
    K = 8 if mode == "welch" else 10
    spe = K * (n // 2 if mode == "welch" else n)
    frames = (28672 * 40960) // spe * K
    cycles = c["GRBM_GUI_ACTIVE"] / 8.0
    out[key] = {
        "bench_flags": flags.strip(), "kernel_s": t, "clock_ghz": cycles / t / 1e9,
        "valu_busy_frac": c["SQ_ACTIVE_INST_VALU"] * 4.0 / (1024.0 * cycles),
        "valu_insts_per_wave_frame": c["SQ_INSTS_VALU"] / (frames * (n // 16) / 64.0),
        "lds_wait_frac_of_wave_cycles": c["SQ_WAIT_INST_LDS"] / c["SQ_WAVE_CYCLES"],
        "lds_array_busy_frac": c["SQ_LDS_IDX_ACTIVE"] / (256.0 * cycles),
        "lds_bank_conflict_frac": c["SQ_LDS_BANK_CONFLICT"] / max(c["SQ_LDS_IDX_ACTIVE"], 1.0),
        "ta_addr_fifo_full_per_wave_cycle": c["SQ_VMEM_TA_ADDR_FIFO_FULL"] / c["SQ_WAVE_CYCLES"],
        "raw": {k: c[k] for k in ("GRBM_GUI_ACTIVE", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_LDS",
                                  "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_DATA_FIFO_FULL",
                                  "SQ_VMEM_TA_ADDR_FIFO_FULL", "SQ_BUSY_CYCLES", "FETCH_SIZE") if k in c},
        "source": f"profiles/valu_counters.json (round {tag}) <- gpurun_out/{sub} (rocprofv3 --pmc, tools/gpu_pmc.sh)",
    }
    print(key, {k: (round(v, 4) if isinstance(v, float) else v) for k, v in out[key].items() if k != "raw"})
json.dump(out, open(os.path.join(R, "profiles", "valu_counters.json"), "w"), indent=1)
