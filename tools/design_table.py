#!/usr/bin/env python3
"""Print DESIGN.md §6's measurement rows from the committed bench lines (profiles/<tag>_bench_*.json), so that the table and
the evidence cannot drift apart.  usage: python tools/design_table.py [tag]"""
import json, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"


def load(name):
    with open(os.path.join(R, "profiles", f"{tag}_bench_{name}.json")) as f:
        return json.loads([ln for ln in f if ln.startswith("{")][-1])


def num(v):
    return f"{v:,.0f}".replace(",", " ")


def row(label, j):
    r = j["roofline"]
    ms = f'{r["kernel_ms_mean"]:.3f} ({r["kernel_ms_min"]:.2f} / {r["kernel_ms_median"]:.2f} / {r["kernel_ms_max"]:.2f})'
    return f'| {label} | {num(j["value"])} | {ms} | {r["achieved"]:.0f} | {100 * r["frac"]:.1f} |'


def alt_row(label, a):
    return f'| {label} | {num(a["Msamples/s"])} | {a["kernel_ms_mean"]:.3f} | {a["GB/s"]:.0f} | {100 * a["frac"]:.1f} |'


rows = []
h = load("headline")
rows.append(row("**headline: 4096-pt energy × 3ch + NF, K=10, 28 672 epochs** — kernel specialised to the reference channel plan (pass 3 keeps 7 of 16 outputs per thread)", h))
try:
    rows.append(row("same batch, no pruning (`--variant 2`, its own run: any other band table, or spectrum output on)", load("headline_unpruned")))
except Exception:
    pass
rows.append(alt_row("the same inside the headline run (`config.alt.unpruned`, 50 launches after the timed region)", h["config"]["alt"]["unpruned"]))
rows.append(alt_row("same kernel, SURVEY.md §8(d)'s 2 GiB batch (`config.alt.cfgH_2GiB_batch`, 6 553 epochs), launches on one stream", h["config"]["alt"]["cfgH_2GiB_batch"]))
if "cfgH_2GiB_batch_two_streams" in h["config"]["alt"]:
    t = h["config"]["alt"]["cfgH_2GiB_batch_two_streams"]
    rows.append(f'| the same 2 GiB batches launched alternately on two streams (`config.alt.cfgH_2GiB_batch_two_streams`: one launch\'s ramp and partly filled last round overlap its neighbour; span of 60 launches / 60) | {num(t["Msamples/s"])} | {t["ms_per_launch"]:.3f} per launch | {t["GB/s"]:.0f} | {100 * t["frac"]:.1f} |')
try:
    rows.append(row("same batch, every twiddle in registers at 3 workgroups per CU (`--variant 23`)", load("headline_all_twiddles_in_registers")))
except Exception:
    pass
if "adc16_input" in h["config"]["alt"]:
    rows.append(alt_row("same batch and kernel, every sample rounded to the USRP's 16-bit wire format (`config.alt.adc16_input`: what the reference's radios deliver; §8)", h["config"]["alt"]["adc16_input"]))
if "wire_format_sc16" in h["config"]["alt"]:
    rows.append(alt_row("**not the headline configuration**: the same samples held in HBM in the radio's wire format, int16 pairs = 4 B per sample (`config.alt.wire_format_sc16`, `crn_sense_run_device_sc16`; outputs bit-identical; GB/s and % are of the 4 B per sample actually read — the kernel is bound by the vector unit at the power cap, §5)", h["config"]["alt"]["wire_format_sc16"]))
rows.append(row("same batch, epoch close reduced to an accumulator reset (`--variant 16`, ablation: not a sensing result)", load("ablation_no_epoch_close")))
rows.append(row("cfg1: 1024-pt energy, 114 688 epochs", load("cfg1_1024pt")))
rows.append(row("2048-pt energy, 57 344 epochs", load("energy_2048pt")))
rows.append(row("512-pt energy, 229 376 epochs", load("energy_512pt")))
rows.append(row("cfg3: 512-pt reference-exact + ANN, 229 376 epochs", load("cfg3_ref512")))
rows.append(row("cfg2: 4096-pt Welch × 64 bands, K = 8, 71 680 epochs (unique bytes; VALU-issue-bound at 1400 W, §5)", load("cfg2_welch")))
rows.append(row("cfg2 at K = 32", load("cfg2_welch_K32")))
rows.append(row("cfg4's N > 1 path on one GPU (`--mode scan --force-collective`: RCCL group of one rank through `crn_comm_*`, all-gather every step)", load("cfg4_scan_one_rank_rccl")))
print("\n".join(rows))
if "--write" in sys.argv:   # replace the GPU rows of DESIGN.md §6 (header .. the first CPU row) in place
    path = os.path.join(R, "DESIGN.md")
    lines = open(path).read().split("\n")
    a = next(i for i, ln in enumerate(lines) if ln.startswith("| Workload (8.75 GiB")) + 2
    lines[a - 2] = lines[a - 2].replace("profiles/r02_bench_", f"profiles/{tag}_bench_")
    b = next(i for i, ln in enumerate(lines) if ln.startswith("| CPU oracle"))
    lines[a:b] = rows
    open(path, "w").write("\n".join(lines))
cb = h["cpu_baseline"]
print(f'CPU: one thread {cb["one_thread"]["value"]:.0f}, {cb["cores"]} threads {cb["value"]:.0f} Msamples/s')
