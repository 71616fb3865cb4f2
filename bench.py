#!/usr/bin/env python3
"""bench.py — throughput of the sensing hot path on MI355X, BASELINE.json's metric.

A "step" is one pass of the hot path (one kernel launch) over one batch of synthetic IQ that is
already resident in HBM: `--epochs` decision epochs of K = 10 frames of N-point complex fp32.
Default workload = the configuration the metric is quoted on (SURVEY.md §8d "cfgH"):
4096-point FFT + energy detect, 3 channels + noise-floor band, threshold decision, 28 672 epochs
= 286 720 frames = 8.75 GiB of IQ per launch (sized for 288 GB of HBM; 35x the Infinity Cache).

  python bench.py                     # 1 GPU
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N        # N GPUs, one rank each
  python bench.py --gpus N            # the same: without a launcher (no WORLD_SIZE) bench.py starts that command itself,
                                      # before it touches the GPU, and relays rank 0's line and the exit code

Multi-GPU: the path shards by stream — every rank owns its own epochs (weak
scaling, per-GPU work fixed) and the only exchange is an RCCL all-gather of the per-epoch
occupancy vector after each step, made through the C ABI (crn_comm_*: side stream, two slots, so it
overlaps the next launch).  torch.distributed (gloo) is the control plane only: the RCCL unique id,
the barriers around the timed region and the max over ranks.

Warm-up: W untimed steps, topped up to ~50 ms of launches when W is small (clock ramp), then
exactly K timed steps between barrier + synchronize pairs, every launch bracketed by its own event
pair on the launch stream.

Prints ONE JSON line on rank 0 (contract in the task statement), including
  roofline      HBM-bound: algorithmic bytes (8 B per unique input sample) / mean kernel time from the
                per-launch events, against the 8 TB/s peak; min / median / max per launch alongside
  roofline_valu (windowed / Welch modes, which are VALU-issue-bound) algorithmic flops against the fp32
                vector peak, plus the measured VALU-busy fraction at the measured clock from the committed
                counter passes of the same command (profiles/valu_counters.json)
  config.alt    (N = 1, headline workload) the same kernel on SURVEY.md §8(d)'s 2 GiB batch, and the
                kernel any other band table / a spectrum request gets (no row pruning)
  cpu_baseline  the oracle (CPU restatement of the reference path, built -O2 -march=native on this box)
                timed on this box's host cores on a bounded sample of the same data: the threads the process may run at once (`value`)
                and one thread — the reference's own topology — BASELINE.md §3's protocol: 2 warm-up passes, median of 10.  At N > 1
                rank 0 measures it after the timed region while the other ranks wait, so every 1 / 2 / 4 / 8-GPU line carries it.

  config        batch_GiB, pruned_rows ("7/16": the kernel keeps the pass-3 rows the reference channel plan reaches; null = none pruned),
                cfgH_as_worded_frac (= alt.cfgH_2GiB_batch.frac): the headline's conditions as fields; `workload` stays under 120 characters

N > 1 is fail-fast: the gloo control plane and a per-stage watchdog on every rank share one limit (--stage-timeout, 300 s): a rank that
dies or hangs — in gloo, inside RCCL, or waiting for the GPU — takes the job down with a non-zero exit within that time instead of
holding the launcher until its own timeout.  Every rank that fails says where in ONE JSON object on stderr (fail_report):
{"bench_failure": {"failed_stage", "rank", "world", "elapsed_s", "why", "rccl": crn_comm_info or null, "visible_devices", "pci_bus_id", "device"}}.
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "cognitive-radio-network_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0       # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)
FP32_VECTOR_PEAK_TF = 157.3  # same guide: peak FP32 (vector), spec, at 2.4 GHz


def usable_cores():
    """CPU threads this process can really run at once: the affinity mask, capped by the cgroup CPU quota
    (the GPU boxes expose 256 hardware threads under a 16-CPU quota: 256 OpenMP threads there run 5x SLOWER
    than 16, tools/cpu_scaling.py)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    why = f"{n} threads = affinity mask"
    for path in ("/sys/fs/cgroup/cpu.max",):
        try:
            quota, period = open(path).read().split()[:2]
            if quota != "max":
                q = max(1, int(math.ceil(int(quota) / int(period))))
                if q < n:
                    n, why = q, f"{q} threads = cgroup CPU quota ({quota.strip()}/{period.strip()})"
        except Exception:
            pass
    try:  # cgroup v1
        q, p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()), int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0 and int(math.ceil(q / p)) < n:
            n, why = int(math.ceil(q / p)), f"{int(math.ceil(q / p))} threads = cgroup CPU quota ({q}/{p})"
    except Exception:
        pass
    return n, why


# What a failing rank at N > 1 says about itself (fail_report): filled in as main() goes along, readable from the watchdog thread
# without any GPU call (a hung main thread may be inside one).
RUN = {"stage": "start", "rank": int(os.environ.get("RANK", "0")), "world": int(os.environ.get("WORLD_SIZE", "1")), "t0": time.monotonic(),
       "multi": False, "rccl": None, "pci_bus_id": None, "device": None, "devices_seen": None}


def visible_devices():
    """What the process was allowed to see: the isolation variables a launcher may have set, and how many devices HIP then shows."""
    v = {k: os.environ[k] for k in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "GPU_DEVICE_ORDINAL") if k in os.environ}
    v["device_count"] = RUN["devices_seen"]
    v["HSA_ENABLE_IPC_MODE_LEGACY"] = os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")
    return v


def fail_report(why, stage=None):
    """ONE JSON object on stderr per failing rank of an N > 1 job (or of the one-GPU rehearsal of that path), so that the launcher's
    tail names the cause — device isolation, RCCL bootstrap, IPC mode, a hung peer — the first time an 8-GPU node runs this:
    {"bench_failure": {"failed_stage", "rank", "world", "elapsed_s", "why", "rccl": crn_comm_info or null, "visible_devices",
    "pci_bus_id", "device"}}.  Uses only what main() cached: no GPU call is made here."""
    if not (RUN["world"] > 1 or RUN["multi"]) or RUN.get("reported"):
        return
    RUN["reported"] = True   # one object per rank: a gloo timeout on the main thread and the watchdog can fire within the same second
    obj = {"bench_failure": {"failed_stage": stage or RUN["stage"], "rank": RUN["rank"], "world": RUN["world"],
                             "elapsed_s": round(time.monotonic() - RUN["t0"], 3), "why": str(why)[:600], "rccl": RUN["rccl"],
                             "visible_devices": visible_devices(), "pci_bus_id": RUN["pci_bus_id"], "device": RUN["device"]}}
    try:
        os.write(2, (json.dumps(obj) + "\n").encode())   # one write: lines of several ranks do not interleave
    except OSError:
        pass


def pci_bus_id_of(torch, index):
    """PCI address of HIP device `index` ("0000:c5:00.0") before any communicator exists, from torch's device properties (the same three
    numbers hipDeviceGetPCIBusId prints; crn_comm_info's own string replaces it once RCCL has bound the rank); None when unavailable."""
    try:
        pr = torch.cuda.get_device_properties(index)
        return f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
    except Exception:   # noqa: BLE001
        return None


class Watchdog:
    """N > 1 only: every rank must pass from one stage of the run to the next within `limit` seconds.  gloo's own timeout covers its
    collectives; this covers what it cannot see — a rank stuck inside ncclCommInitRank, or in a device synchronise behind an
    all-gather that a missing peer never completes.  On expiry the rank says where it was and exits 6 (os._exit: the main thread is
    blocked in native code), which makes torch.distributed.run take the other ranks down."""

    def __init__(self, limit, rank):
        import threading
        self.limit, self.rank, self.stage, self.t = float(limit), rank, "start", time.monotonic()
        self.done = False
        threading.Thread(target=self._watch, daemon=True).start()

    def pet(self, stage):
        self.stage, self.t = stage, time.monotonic()
        RUN["stage"] = stage

    def stop(self):
        self.done = True

    def _watch(self):
        while not self.done:
            time.sleep(min(1.0, self.limit / 10))
            if not self.done and time.monotonic() - self.t > self.limit:
                print(f"bench.py: rank {self.rank}: stage '{self.stage}' did not finish within {self.limit:.0f} s (a peer hung or died?): "
                      "exiting 6 so that the launcher stops the job", file=sys.stderr, flush=True)
                fail_report(f"watchdog: stage did not finish within {self.limit:.0f} s (a peer hung or died, or this rank is stuck in "
                            "native code)", self.stage)
                os._exit(6)


def run_ranks(cmd, env):
    """Run the launcher: its stdout is collected (rank 0's JSON line is in it), its stderr goes straight through to ours as it comes — a
    caller that kills a hung job still has everything said so far — while the ranks' `bench_failure` objects are remembered.
    Returns (exit code, stdout text, [failure lines])."""
    import subprocess
    import threading
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, text=True, errors="replace")
    failures = []

    def pump():
        for ln in p.stderr:
            sys.stderr.write(ln)
            sys.stderr.flush()
            if ln.lstrip().startswith('{"bench_failure"'):
                failures.append(ln.strip())
    t = threading.Thread(target=pump, daemon=True)
    t.start()
    out = p.stdout.read()
    rc = p.wait()
    t.join(timeout=10)
    return rc, out, failures


def self_launch(n):
    """`python bench.py --gpus N` started without a launcher: run `python -m torch.distributed.run --nnodes=1 --nproc-per-node N
    --master-addr 127.0.0.1 --master-port <free> bench.py <same arguments>` as a child process (one rank per GPU; when the node
    shows fewer devices than ranks the ranks share device 0, which only a stand-in RCCL behind $CRN_RCCL_LIB accepts), pass on the
    one JSON line rank 0 prints, and return the launcher's exit code.  Called before anything in this process touches the GPU."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [os.path.realpath(sys.executable), "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL between processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "1")              # (what torch.distributed.run would set, without its warning)
    rc, out, failures = run_ranks(cmd, env)
    lines = [ln for ln in out.splitlines() if ln.lstrip().startswith("{")]
    for ln in out.splitlines():
        if ln not in lines and ln.strip():
            print(ln, file=sys.stderr)
    if lines:
        print(lines[-1], flush=True)
    if rc != 0 and failures:
        # the launcher's own summary (a screenful per job) comes after the ranks' one-line accounts: say them again, last, so that
        # whoever keeps only the tail of this process's output still reads the cause
        print(f"bench.py: the job failed (exit {rc}); what the failing rank(s) said, first failure first:", file=sys.stderr)
        for ln in failures[:8]:
            print(ln, file=sys.stderr)
        sys.stderr.flush()
    if rc == 0 and not lines:
        print("bench.py: the ranks exited cleanly but rank 0 printed no JSON line", file=sys.stderr)
        return 1
    return rc


def live_counters(args, epochs, passes):
    """rocprofv3 --pmc passes (one counter group per pass, --kernel-trace only) over a short child run of the same workload:
    {counter: mean over the timed launches, "kernel_s": mean kernel duration of the last pass}, or None."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if shutil.which("rocprofv3") is None:
        return None
    tmp = tempfile.mkdtemp(prefix="crn_pmc_", dir="/tmp")
    # the interpreter running this file, resolved to the real binary: no shim or wrapper script may sit between rocprofv3's `--`
    # and the program (the profiler's preloaded library initialises the GPU, and an exec after that is refused on this pool)
    child = [os.path.realpath(sys.executable), os.path.abspath(__file__), "--steps", "5", "--warmup", "20", "--cpu-epochs", "0", "--no-alt",
             "--no-live-traffic", "--fft", str(args.fft), "--mode", args.mode, "--variant", str(args.variant),
             "--epochs", str(epochs)]
    if args.frames > 0:
        child += ["--frames", str(args.frames)]
    if getattr(args, "wire_format", False):
        child += ["--wire-format"]
    elif getattr(args, "adc_bits", 0):
        child += ["--adc-bits", str(args.adc_bits)] + (["--uhd-scale"] if getattr(args, "uhd_scale", False) else [])
    got = {}
    try:
        for i, group in enumerate(passes):
            out = os.path.join(tmp, str(i))
            r = subprocess.run(["rocprofv3", "--pmc", *group, "--kernel-trace", "--output-format", "csv", "-d", out, "--"] + child,
                               cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL,
                               stderr=subprocess.PIPE, text=True, timeout=180)
            if r.returncode != 0:
                print(f"bench: counter pass {group} failed (rc {r.returncode}); roofline.traffic falls back to the committed figure\n"
                      + r.stderr[-1500:], file=sys.stderr)
                return None
            per = {}
            for f in glob.glob(os.path.join(out, "**", "*_counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if "sense_kernel" in row["Kernel_Name"]:
                        per.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
            for ctr in group:
                if ctr not in per:
                    return None
                got[ctr] = sum(per[ctr][-5:]) / len(per[ctr][-5:])   # the timed launches
            dur = [(int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-9
                   for f in glob.glob(os.path.join(out, "**", "*_kernel_trace.csv"), recursive=True)
                   for row in csv.DictReader(open(f)) if "sense_kernel" in row["Kernel_Name"]][-5:]
            if dur:
                got["kernel_s"] = sum(dur) / len(dur)
    except Exception as e:   # noqa: BLE001 — a measurement aid must not take the bench line down with it
        print(f"bench: counter passes failed: {e!r}", file=sys.stderr)
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return got


def live_traffic(args, epochs):
    """(bytes per launch, how) from FETCH_SIZE and WRITE_SIZE passes (they cannot share a pass on gfx950), or (None, None).
    FETCH_SIZE is in KiB and on gfx950 tallies each 128-B request as 64 B (x2: MI355X_MICROARCH.md, HBM section;
    re-calibrated for this kernel's 8-byte loads in profiles/r01_fetch_size_calibration.txt); WRITE_SIZE reads true."""
    got = live_counters(args, epochs, [["FETCH_SIZE"], ["WRITE_SIZE"]])
    if not got:
        return None, None
    total = int(got["FETCH_SIZE"] * 1024 * 2 + got["WRITE_SIZE"] * 1024)
    return total, ("measured in this run: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes, --kernel-trace only) over a "
                   "5-step child run of this workload; FETCH_SIZE KiB x 1024 x 2 (gfx950 counts 128-B requests as 64 B) + WRITE_SIZE KiB x 1024")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--fft", type=int, default=4096, help="FFT length N (headline: 4096)")
    ap.add_argument("--epochs", type=int, default=0,
                    help="decision epochs per step: per GPU with --scaling weak, in total with --scaling strong (0 = 8.75 GiB of IQ)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak (default): every GPU takes --epochs epochs per step, the job grows with N.  strong: --epochs is the "
                         "whole job (the N = 1 batch), split evenly over the ranks; a share under 4 GiB alternates its launches "
                         "between two streams so that one launch's ramp and drain overlap the next")
    ap.add_argument("--two-streams", action="store_true",
                    help="alternate the launches between two streams whatever the batch size (what --scaling strong does by itself below 4 GiB per GPU)")
    ap.add_argument("--mode", choices=["energy", "ref", "welch", "scan"], default="energy",
                    help="scan = cfg4's wideband scan: the welch kernel, streams of 64 channels sharded over the ranks")
    ap.add_argument("--variant", type=int, default=0, help="kernel variant (0 = default)")
    ap.add_argument("--frames", type=int, default=0, help="frames per epoch K (0 = the configuration's own, 10)")
    ap.add_argument("--cpu-epochs", type=int, default=-1, help="oracle sample size (-1 = auto, 0 = skip the CPU baseline)")
    ap.add_argument("--no-alt", action="store_true", help="skip the config.alt legs (2 GiB batch, unpruned kernel)")
    ap.add_argument("--span-events", action="store_true",
                    help="one event pair around all timed launches instead of one pair per launch")
    ap.add_argument("--no-check", action="store_true", help="skip output checks (ablation variants only)")
    ap.add_argument("--wire-format", action="store_true",
                    help="keep the samples in HBM in the radio's wire format (int16 pairs, 4 bytes per complex sample: "
                         "crn_sense_run_device_sc16) instead of complex floats; implies --adc-bits 16.  Not the headline configuration")
    ap.add_argument("--uhd-scale", action="store_true",
                    help="with --adc-bits 16: scale the integer samples by 1/32767 (the constant UHD's sc16 -> fc32 converter uses) instead "
                         "of the power of two 1/32768: the mantissas fill up (diagnostic)")
    ap.add_argument("--adc-bits", type=int, default=0,
                    help="round every input sample to this many bits (16 = the USRP wire format the reference's radios deliver); "
                         "0 = full-precision fp32 (the default: SURVEY.md §8(d)'s generator, the worst case for power)")
    ap.add_argument("--zeros", action="store_true",
                    help="diagnostic, not a result: all-zero IQ (same instruction stream, least switching energy) — how much "
                         "of the kernel time is the clock the chip holds under load (implies --no-check, no CPU baseline)")
    ap.add_argument("--force-collective", action="store_true",
                    help="dry run of the N>1 code path on one GPU: RCCL group of one rank, all-gather every step")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="do not measure roofline.traffic with rocprofv3 counter passes of a short child run (N = 1 only); "
                         "use the committed figure instead")
    ap.add_argument("--stage-timeout", type=float, default=300.0,
                    help="N > 1: seconds a rank may spend in one stage of the run (gloo collectives and the watchdog): a hung or dead "
                         "peer ends the job with a non-zero exit within this time")
    ap.add_argument("--traffic-json", default=os.path.join(ROOT, "profiles", "hbm_traffic.json"))
    ap.add_argument("--valu-json", default=os.path.join(ROOT, "profiles", "valu_counters.json"))
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start the N rank processes ourselves.  This process has made no GPU
        # call (torch is not even imported yet) and never will: it only relays rank 0's JSON line and the launcher's exit code.
        raise SystemExit(self_launch(args.gpus))

    # stdout carries exactly one JSON line: everything else this process (or a library under it:
    # RCCL prints a version banner at init) writes to fd 1 goes to stderr instead.
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    # ranks started by a launcher that did not export it: RCCL between processes needs dmabuf IPC on this driver, and the HSA
    # runtime reads the variable when the first HIP call initialises it (nothing has touched the GPU yet)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    import numpy as np
    import torch
    import torch.distributed as dist

    if args.variant not in (0, 2, 13) and args.variant < 100:
        os.environ["CRN_SENSE_AB"] = "1"   # measurement variants live in libcrnsense_ab.so, not in the shipped library
    if args.wire_format and "CRN_SENSE_LIB" not in os.environ:
        # the wire-format kernels are optional: libcrnsense_sc16.so (make -C csrc SC16=1), not the shipped library
        os.environ["CRN_SENSE_LIB"] = os.path.join(ROOT, "cognitive-radio-network_amd", "libcrnsense_sc16.so")
    import crnsense as cs
    from sharding import make_device_exchange, shard

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: libcrnsense has no CPU path")
    if local_rank >= torch.cuda.device_count():   # a launcher that shows each rank only its own GPU
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    multi = world > 1 or args.force_collective
    RUN.update(rank=rank, world=world, multi=multi, device=local_rank, devices_seen=torch.cuda.device_count(),
               pci_bus_id=pci_bus_id_of(torch, local_rank))
    dog = None
    if world > 1:
        import datetime
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        dog = Watchdog(args.stage_timeout, rank)
        # control plane only; the data-path collective is crn_comm_* (RCCL).  gloo's default timeout is 30 minutes — the driver's whole
        # budget for a scaling run: one wedged rank must cost --stage-timeout, not the run
        dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=args.stage_timeout))

    def stage(name):
        RUN["stage"] = name
        if dog is not None:
            dog.pet(name)
    stage("set-up")

    if args.mode == "ref":
        cfg = cs.cfg_reference()
        workload = "cfg3: 512-pt |X| mean x10 + 4-5-3 ANN cascade (reference-exact)"
        label = "512-pt x 3ch reference-exact + ANN"
    elif args.mode in ("welch", "scan"):
        cfg = cs.cfg_welch(args.fft, 8, 64)
        # per-band threshold = lambda x the noise-floor estimate (lambda = 4): a band of N/64 bins of
        # Hann-windowed noise of power 1e-6 holds (N/64) x N x 1e-6 x mean(w^2), mean(w^2) = 3/8
        for b in range(64):
            cfg.thresh[b] = 4.0 * (args.fft / 64) * args.fft * 1e-6 * 0.375
        workload = f"cfg2: {args.fft}-pt Welch PSD (Hann, 50% overlap) x 64 bands + threshold"
        label = f"{args.fft}-pt Welch x 64ch"
        if args.mode == "scan":
            # BASELINE.json configs[4]: 256 channels = 4 streams x 64 bands is the shard unit; a rank's batch
            # is thousands of such units (one epoch of one stream each), the occupancy gather is [epochs, 64]
            workload = (f"cfg4: wideband scan, {args.fft}-pt Welch x 64 channels per stream (256 channels = 4 streams), "
                        "streams sharded over the ranks, occupancy all-gather")
            label = f"{args.fft}-pt Welch scan x 64ch per stream"
    else:
        cfg = cs.cfg_energy_scaled(args.fft, 4.0)
        workload = f"{args.fft}-pt FFT + energy detect x 3ch (+noise-floor band), K=10, threshold"
        label = f"{args.fft}-pt x 3ch"
    cfg.device = local_rank
    if args.frames > 0:
        cfg.frames_per_epoch = args.frames
        workload += f" [K={args.frames}]"
    N, K = cfg.fft_len, cfg.frames_per_epoch
    spe = cs.samples_per_epoch(cfg)
    E_arg = args.epochs if args.epochs > 0 else (28672 * 40960) // spe
    # weak scaling: E_arg epochs on every GPU.  strong scaling: E_arg epochs in all — the N = 1 batch — dealt evenly (the exchange
    # carries equal blocks: a remainder of < N epochs is left out and the job size printed is what ran)
    E = E_arg if args.scaling == "weak" else max(1, E_arg // world)
    n_samples = cs.samples_needed(cfg, E)
    lo, hi = shard(E * world, rank, world)
    assert hi - lo == E
    # a small share per GPU: ramp + drain are > 9 % of a 1 GiB launch (measured with in-kernel time stamps in round 3); two streams overlap them
    two_streams = args.two_streams or (args.scaling == "strong" and E * spe * 8 < (4 << 30))

    sensor = cs.Sensor(cfg)
    sensor.set_variant(args.variant)
    info = sensor.kernel_info()
    if os.environ.get("CRN_SENSE_AB") == "1":
        workload += " [A/B BUILD libcrnsense_ab.so: measurement variant, not the shipped library]"
    elif os.path.basename(cs.LIB_PATH) != "libcrnsense.so":
        workload += f" [LIBRARY {os.path.basename(cs.LIB_PATH)}, not the shipped libcrnsense.so]"
    pruned = "PASS3_ROWS" in info["name"]
    workload_note = None
    if pruned:
        n_kept = info["name"].split("PASS3_ROWS=")[1].split("-of-16")[0]
        # (short, so that the whole headline workload string stays under the 120 characters the driver's record keeps; the long form
        # rides in config.workload_note)
        workload += f" [pruned: {n_kept}/16 pass-3 rows, ref. channel plan]"
        workload_note = (f"kernel specialised to the reference channel plan: pass 3 and the accumulate keep {n_kept} of 16 outputs per thread "
                         "(outputs bit-identical, every input byte still read); --variant 2 / config.alt.unpruned = what any other band "
                         "plan or a spectrum request runs")

    iq = torch.zeros(n_samples * 2, dtype=torch.float32, device=dev)
    truth = torch.empty(E, dtype=torch.int32, device=dev)
    n_sets = 2 if two_streams else 1      # launches in flight at once never share an output buffer: one set per stream
    sets = [{"feats": torch.empty(E, cfg.n_bands, dtype=torch.float32, device=dev), "occ": torch.empty(E, cfg.n_bands, dtype=torch.uint8, device=dev),
             "dec": torch.empty(E, dtype=torch.int32, device=dev), "ann": torch.empty(E, 3, dtype=torch.float64, device=dev)} for _ in range(n_sets)]
    feats, occ, dec, ann = (sets[0][k] for k in ("feats", "occ", "dec", "ann"))
    stream = torch.cuda.current_stream().cuda_stream
    tstreams = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)] if two_streams else [torch.cuda.current_stream()]
    if args.zeros:
        args.no_check, args.cpu_epochs, args.no_alt = True, 0, True
        truth.zero_()
        workload += " [DIAGNOSTIC: all-zero input]"
    else:
        if args.wire_format:
            args.adc_bits, args.no_alt = 16, True
        sc0 = cs.SynthCfg()
        sc0.seed, sc0.noise_power, sc0.signal_rms, sc0.tones_per_band = 0xC0FFEE + 1000 * rank, 1e-6, 0.02, 8
        sc0.pu_model, sc0.signal_kind, sc0.n_streams, sc0.adc_bits = cs.PU_UNIFORM, cs.SIG_TONES, 1, args.adc_bits
        sensor.synth_fill_device_ex(iq.data_ptr(), E, spe, sc0, truth_ptr=truth.data_ptr(), stream=stream)
        if args.adc_bits:
            workload += f" [DIAGNOSTIC INPUT: samples rounded to {args.adc_bits} bits, as a radio delivers them]"
            if args.uhd_scale and not args.wire_format:
                iq.mul_(32768.0 / 32767.0)
                workload += " [scaled by 1/32767 instead of 1/32768]"
    src, sample_bytes = iq, 8
    if args.wire_format and not args.zeros:
        wire = torch.empty(n_samples * 2, dtype=torch.int16, device=dev)
        sensor.pack_sc16_device(iq.data_ptr(), n_samples, wire.data_ptr(), stream=stream)
        src, sample_bytes = wire, 4
        workload += " [NOT THE HEADLINE CONFIGURATION: samples held in HBM in the radio's wire format, int16 pairs = 4 B per sample, converted in the kernel's first pass; outputs bit-identical to the float path]"
    out_sets = [{"features": t["feats"].data_ptr(), "ann_out": t["ann"].data_ptr(), "decision": t["dec"].data_ptr(),
                 "occupancy": t["occ"].data_ptr(), "spectrum": 0} for t in sets]
    outs = out_sets[0]
    noise_floor = None
    if args.mode in ("welch", "scan") and not args.zeros:
        # SURVEY.md §8(d) cfg2 as worded: thr_b = lambda x NF_est, NF_est = the median band energy — measured on this rank's own
        # batch (one untimed pass), not assumed: crn_noise_floor_device, then crn_sense_set_thresholds (which also updates `cfg`,
        # so the oracle check below compares against the same f32 thresholds)
        analytic = (args.fft / 64) * args.fft * 1e-6 * 0.375
        sensor.run_device(src.data_ptr(), E, N, outs, stream=stream, sc16=sample_bytes == 4)
        nf = sensor.noise_floor(feats.data_ptr(), E, stream=stream)
        if not (0.8 * analytic < nf < 1.25 * analytic):
            raise SystemExit(f"bench: noise-floor estimate {nf:.4g} is not near the generator's {analytic:.4g}")
        sensor.set_thresholds([float(np.float32(4.0) * np.float32(nf))] * cfg.n_bands, stream=stream)
        noise_floor = {"estimate_median_band_energy": nf, "generator_expectation": analytic, "lambda": 4.0}
    # N > 1: the occupancy block alternates between two slots of the C ABI's communicator so that the
    # all-gather of step i (side stream) overlaps the sensing kernel of step i + 1
    # (two streams: four slots, so that a stream's next launch never waits for the gather of its previous one)
    stage("communicator (ncclCommInitRank is collective)")
    ex, ex_kind = make_device_exchange(E, cfg.n_bands, local_rank, rank, world, depth=4 if two_streams else 2) if multi else (None, "")
    if ex is not None:
        RUN["rccl"] = ex.info()   # (cached: a failure report must not call into RCCL from beside a hung main thread)
        RUN["pci_bus_id"] = RUN["rccl"].get("pci_bus_id") or RUN["pci_bus_id"]
    torch.cuda.synchronize()   # the set-up above ran on the current stream; the timed launches may run on others
    stage("warm-up")
    n_done = 0

    def step(sn, epochs, ev=None):
        """One pass of the hot path over this rank's batch: launch i goes to stream i mod (1 or 2) with that stream's output set."""
        nonlocal n_done
        ts, out_ptrs = tstreams[n_done % len(tstreams)], out_sets[n_done % n_sets]
        st = ts.cuda_stream
        if ex is not None:
            out_ptrs["occupancy"] = ex.local_ptr(n_done, st)
        if ev is not None:
            ev[0].record(ts)
        sn.run_device(src.data_ptr(), epochs, N, out_ptrs, stream=st, sc16=sample_bytes == 4)
        if ev is not None:
            ev[1].record(ts)
        if ex is not None:
            ex.exchange(n_done, st)
        n_done += 1

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    # Clock ramp: on this part a cold process needs ~25 ms of back-to-back launches before the
    # per-launch time settles, so the W warm-up steps are topped up to at least
    # ~50 ms of untimed work when W is small.  The timed region below is exactly K steps.
    prewarm = max(args.warmup, int(0.05 / 1.6e-3 * (28672 * 40960) / max(E * spe, 1)) + 1)
    for _ in range(prewarm + (prewarm + n_done) % len(tstreams)):   # (the timed region starts on stream 0)
        step(sensor, E)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    span0 = torch.cuda.Event(enable_timing=True)
    span1 = [torch.cuda.Event(enable_timing=True) for _ in tstreams]

    stage("timed region")
    barrier()
    t0 = time.perf_counter()
    span0.record(tstreams[0])
    for ts in tstreams[1:]:
        ts.wait_event(span0)
    for i in range(args.steps):
        step(sensor, E, None if args.span_events else ev[i])
    for ts, e1 in zip(tstreams, span1):
        if ex is not None:
            ex.finish(ts.cuda_stream)
        e1.record(ts)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    # Kernel duration from events on the launch stream: one pair per launch (default), so min / median /
    # max are real per-launch figures; --span-events: one pair around the K back-to-back launches.
    # Two streams: launches overlap, so a launch's own event pair also spans part of its neighbour — the duration that
    # prices the roofline is then the span from the first launch to the last event on either stream, per launch.
    span_ms = max(span0.elapsed_time(e1) for e1 in span1)
    per_launch_ms = None if args.span_events else [a.elapsed_time(b) for a, b in ev]
    if args.span_events or two_streams:
        kern_ms = [span_ms / args.steps]
    else:
        kern_ms = per_launch_ms
    kern_ms_mean = float(np.mean(kern_ms))
    samples_per_step = E * spe * world
    value = samples_per_step * args.steps / dt / 1e6  # Msamples/s, whole job
    algo_bytes = E * spe * sample_bytes               # per launch: 8 B per unique input sample (4 in wire format)
    achieved = algo_bytes / (kern_ms_mean * 1e-3) / 1e9

    # ---- sanity on the timed outputs ----------------------------------------------------------------
    stage("output checks")
    last = n_done - 1
    feats, occ, dec, ann = (sets[last % n_sets][k] for k in ("feats", "occ", "dec", "ann"))   # what the last timed launch wrote
    occ_host = ex.local_host(last) if ex is not None else occ.cpu().numpy()
    picked = truth.cpu().numpy()
    if args.no_check:
        pass
    elif cfg.decide == cs.DECIDE_THRESHOLD and cfg.ref_band >= 0:
        want = np.zeros_like(occ_host)
        idx = np.nonzero(picked > 0)[0]
        want[idx, picked[idx]] = 1
        mism = int((occ_host != want).any(axis=1).sum())
        if mism:
            raise SystemExit(f"bench: {mism} epochs whose occupancy differs from the driven pattern")
    elif cfg.decide == cs.DECIDE_THRESHOLD:
        # Welch / scan: the driven channel must read occupied in every epoch (others may too: an epoch's last frame
        # reaches half a frame into the next epoch's traffic, and a tone cut off mid-frame splatters)
        idx = np.nonzero(picked > 0)[0]
        mism = int((occ_host[idx, picked[idx] - 1] != 1).sum())
        if mism:
            raise SystemExit(f"bench: {mism} epochs whose driven channel does not read occupied")
    elif cfg.decide == cs.DECIDE_ANN:
        mism = int((dec.cpu().numpy() != picked).sum())
        if mism:
            raise SystemExit(f"bench: {mism} epochs whose decision differs from the driven pattern")
    if not args.no_check and rank == 0:
        # every mode, even with --cpu-epochs 0: a small sample of the timed launch's outputs against the oracle
        import oracle_py as orc
        n_chk = min(E, 16)
        host_iq = iq[: cs.samples_needed(cfg, n_chk) * 2].cpu().numpy()
        ref = orc.run(cfg, host_iq, n_chk)
        g = feats[:n_chk].cpu().numpy()
        rel = np.abs(g - ref["features"]) / np.maximum(np.abs(ref["features"]), 1e-30)
        if rel.max() > 1e-5 or not np.array_equal(occ_host[:n_chk], ref["occupancy"]):
            raise SystemExit(f"bench: GPU results differ from the oracle on the first {n_chk} epochs (rel {rel.max():.3g})")
    if ex is not None and not args.no_check:
        # every rank must find its own block, unchanged, at its place in the gathered vector
        torch.cuda.synchronize()
        if not np.array_equal(ex.gathered_host(last)[rank * E:(rank + 1) * E], occ_host):
            raise SystemExit(f"bench: rank {rank}: gathered occupancy differs from the local block")

    def committed(path, key):
        if not os.path.exists(path):
            return None
        try:
            return json.load(open(path)).get(key)
        except Exception:
            return None

    mode_key = f"{'welch' if args.mode == 'scan' else args.mode}{N}"
    traffic, traffic_source = None, None
    profiled = "rocprof" in os.environ.get("LD_PRELOAD", "") or "ROCP_TOOL_LIBRARIES" in os.environ   # already under a profiler
    if rank == 0 and world == 1 and not multi and not args.no_live_traffic and not args.zeros and not profiled:
        # HBM bytes per launch, measured now: two rocprofv3 counter passes (FETCH_SIZE and WRITE_SIZE cannot share a
        # pass on gfx950) over a short child run of this same workload — a child process, started while this one idles.
        traffic, traffic_source = live_traffic(args, E)
    if traffic is None and not args.wire_format:   # (the committed figures are for complex-float input)
        tj = committed(args.traffic_json, mode_key)
        if tj:  # measured once per kernel with rocprofv3 PMC passes of this command; scales linearly with the batch
            traffic = int(tj["hbm_bytes_per_launch"] * (E / tj["epochs"]))
            traffic_source = (f"committed ({os.path.relpath(args.traffic_json, ROOT)}: rocprofv3 FETCH_SIZE x2 + WRITE_SIZE passes of this "
                              "command at N = 1), scaled by the batch" + (", not re-measured at N > 1" if world > 1 else ""))

    # ---- second roofline for the windowed kernels: VALU ---------------------------------------------
    roofline_valu = None
    if cfg.window != cs.WINDOW_RECT or args.mode in ("welch", "scan"):
        frames = E * K
        # algorithmic flops per N-point frame: FFT 5 N log2 N, window 2 N, |X|^2 accumulate 4 N
        flops = frames * (5.0 * N * math.log2(N) + 2.0 * N + 4.0 * N)
        tf = flops / (kern_ms_mean * 1e-3) / 1e12
        roofline_valu = {"bound": "valu", "achieved": tf, "peak": FP32_VECTOR_PEAK_TF, "unit": "TFLOP/s",
                         "frac": tf / FP32_VECTOR_PEAK_TF,
                         "note": "algorithmic flops (5 N log2 N + 6 N per frame; every hop of N/2 new samples costs a whole "
                                 "N-point transform) against the fp32 vector peak at 2.4 GHz; butterflies are add-heavy, "
                                 "so issue slots, not flops, are what runs out: see issue_frac"}
        lv = None
        if rank == 0 and world == 1 and not multi and not args.no_live_traffic and not args.zeros and not profiled:
            lv = live_counters(args, E, [["SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU", "GRBM_GUI_ACTIVE"]])
        if lv and "kernel_s" in lv:
            cycles = lv["GRBM_GUI_ACTIVE"] / 8.0                      # summed over the 8 XCDs
            roofline_valu.update({"issue_frac": lv["SQ_ACTIVE_INST_VALU"] * 4.0 / (1024.0 * cycles),   # quad-cycles, 1024 SIMDs
                                  "clock_ghz": cycles / lv["kernel_s"] / 1e9,
                                  "valu_insts_per_wave_frame": lv["SQ_INSTS_VALU"] / (frames * (N // 16) / 64.0),
                                  "issue_frac_source": "measured in this run: rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU "
                                                       "GRBM_GUI_ACTIVE over a 5-step child run; clock = GRBM_GUI_ACTIVE / 8 / kernel time"})
        else:
            vj = committed(args.valu_json, mode_key)
            if vj:
                roofline_valu.update({"issue_frac": vj["valu_busy_frac"], "clock_ghz": vj["clock_ghz"],
                                      "valu_insts_per_wave_frame": vj["valu_insts_per_wave_frame"],
                                      "issue_frac_source": vj["source"] + " (committed)"})

    # ---- config.alt: the same metric on SURVEY.md §8(d)'s 2 GiB batch, and without row pruning -------
    alt = None
    if rank == 0 and world == 1 and not multi and not args.no_alt and args.mode == "energy" and args.variant == 0 \
            and args.epochs == 0 and args.frames == 0 and args.adc_bits == 0:
        def leg(sn, epochs, n=50, src=None, dst=None):
            src = iq if src is None else src
            dst = outs if dst is None else dst
            for _ in range(max(20, int(0.03 / 1.6e-3 * E / epochs))):
                sn.run_device(src.data_ptr(), epochs, N, dst, stream=stream)
            pairs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
            for a, b in pairs:
                a.record()
                sn.run_device(src.data_ptr(), epochs, N, dst, stream=stream)
                b.record()
            torch.cuda.synchronize()
            ms = [a.elapsed_time(b) for a, b in pairs]
            gbs = epochs * spe * 8 / (float(np.mean(ms)) * 1e-3) / 1e9
            return {"epochs": epochs, "bytes_per_step": epochs * spe * 8, "kernel_ms_mean": float(np.mean(ms)),
                    "kernel_ms_median": float(np.median(ms)), "kernel_ms_min": float(np.min(ms)), "GB/s": gbs, "frac": gbs / HBM_PEAK_GBS,
                    "Msamples/s": epochs * spe / (float(np.mean(ms)) * 1e-3) / 1e6}

        def leg_two_streams(sn, epochs, n=60):
            """Consecutive batches launched alternately on two streams: a launch's ramp and its last, partly filled round of
            workgroups overlap the neighbouring launch instead of leaving the machine part empty (launches on ONE stream
            serialise: the next kernel waits for the previous one's last workgroup).  Batches are independent — no state crosses
            epochs — so this is how a caller with a queue of batches runs them.  Timed as a span: first launch to the last
            event on either stream; 4 disjoint output sets so that launches in flight never share a buffer."""
            sa, sb = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
            sets = []
            for _ in range(4):
                t = [torch.empty(epochs, cfg.n_bands, dtype=torch.float32, device=dev), torch.empty(epochs, 3, dtype=torch.float64, device=dev),
                     torch.empty(epochs, dtype=torch.int32, device=dev), torch.empty(epochs, cfg.n_bands, dtype=torch.uint8, device=dev)]
                sets.append((t, {"features": t[0].data_ptr(), "ann_out": t[1].data_ptr(), "decision": t[2].data_ptr(),
                                 "occupancy": t[3].data_ptr(), "spectrum": 0}))
            torch.cuda.synchronize()

            def burst(count):
                for i in range(count):
                    st = (sa, sb)[i & 1]
                    sn.run_device(iq.data_ptr(), epochs, N, sets[i & 3][1], stream=st.cuda_stream)
            burst(max(20, int(0.03 / 1.6e-3 * E / epochs)))
            torch.cuda.synchronize()
            e0, ea, eb = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            e0.record(sa)
            sb.wait_event(e0)
            burst(n)
            ea.record(sa)
            eb.record(sb)
            torch.cuda.synchronize()
            span_ms = max(e0.elapsed_time(ea), e0.elapsed_time(eb))
            same = all(torch.equal(a, b) for a, b in zip(sets[0][0], sets[1][0]))   # same input, same kernel: identical outputs
            gbs = epochs * spe * 8 * n / (span_ms * 1e-3) / 1e9
            return {"epochs": epochs, "bytes_per_step": epochs * spe * 8, "launches": n, "span_ms": span_ms, "ms_per_launch": span_ms / n,
                    "GB/s": gbs, "frac": gbs / HBM_PEAK_GBS, "Msamples/s": epochs * spe * n / (span_ms * 1e-3) / 1e6,
                    "outputs_identical_across_streams": bool(same)}
        e2g = (2 ** 28) // spe   # SURVEY.md §8(d) cfgH: B = 2^28 / 4096 = 65 536 frames = 2 GiB
        alt = {"cfgH_2GiB_batch": dict(leg(sensor, e2g), kernel=info["name"][:40] + "...",
                                       note="SURVEY.md §8(d) cfgH batch: 2^28 samples per launch (launch ramp and tail weigh more); launches "
                                            "on one stream, one event pair per launch")}
        alt["cfgH_2GiB_batch_two_streams"] = dict(leg_two_streams(sensor, e2g), kernel=info["name"][:40] + "...",
                                                  note="the same 2 GiB batches launched alternately on two streams, so that one launch's "
                                                       "ramp and partly filled last round overlap its neighbour (span over 60 launches / 60)")
        s2 = cs.Sensor(cfg)
        s2.set_variant(2)
        alt["unpruned"] = dict(leg(s2, E), kernel=s2.kernel_info()["name"],
                               note="what any band table outside the reference plan's rows, or a spectrum request, runs")
        s2.close()

    # ---- CPU baseline ---------------------------------------------------------------------------------
    # north_star: "throughput is reported at 1/2/4/8 GPUs next to the liquid-dsp CPU path timed on the same box's host cores (count
    # stated)" — so at every N: rank 0 measures after the timed region, the other ranks wait at the barrier below (idle: the host cores
    # are rank 0's).  Protocol = BASELINE.md §3: 2 warm-up passes, median of 10 timed passes; topology A = one thread (the reference's
    # single CE pthread, src/extensible_cognitive_radio.cpp:1761-1808), topology B = the threads this process may run at once.
    stage("cpu baseline (rank 0 measures, the others wait)")
    cpu = None
    if rank == 0 and args.cpu_epochs != 0:
        import oracle_py as orc
        native = orc.use_native_build()       # -O2 -march=native, compiled on this box (the portable build otherwise)
        hw = os.cpu_count() or 1
        quota, why = usable_cores()
        cap = max(1, (7168 * 40960) // spe)   # at most 2.2 GiB of the batch goes to the host
        n_all = args.cpu_epochs if args.cpu_epochs > 0 else min(E, cap)
        host_iq = iq[: cs.samples_needed(cfg, n_all) * 2].cpu().numpy()
        quick = args.cpu_epochs > 0           # explicit sample size: one warm-up, two passes (tests)
        n_warm, n_pass = (1, 2) if quick else (2, 10)

        def timed(n_ep, threads, min_s):
            """(median rate, rates, repetitions per pass, last outputs): n_warm untimed passes, then n_pass timed ones of >= min_s each."""
            t1 = time.perf_counter()
            ref_out = orc.run(cfg, host_iq, n_ep, n_threads=threads)            # warm-up 1 (pages in the sample, sizes the passes)
            one = time.perf_counter() - t1
            reps = 1 if quick else max(1, int(math.ceil(min_s / max(one, 1e-3))))
            for _ in range(n_warm - 1):
                for _ in range(reps):
                    orc.run(cfg, host_iq, n_ep, n_threads=threads)
            rates = []
            for _ in range(n_pass):
                t1 = time.perf_counter()
                for _ in range(reps):
                    ref_out = orc.run(cfg, host_iq, n_ep, n_threads=threads)
                rates.append(n_ep * spe * reps / (time.perf_counter() - t1) / 1e6)
            return float(np.median(rates)), rates, reps, ref_out

        # "all host cores" = the threads this process may really run at once: the affinity mask capped by the cgroup CPU quota (the GPU
        # boxes expose 256 hardware threads under a 16-CPU quota).  That count is what `value` is measured on — not the best of a
        # sweep; a short sustained pass at 2x and 4x the quota is recorded beside it for information only (bursts above the quota run
        # for a fraction of a second, tools/cpu_scaling.py).
        cores, sweep = quota, {}
        if not quick:
            for th in sorted({c for c in (2 * quota, 4 * quota) if 1 <= c <= hw}):
                orc.run(cfg, host_iq, min(n_all, 2 * th), n_threads=th)
                t1 = time.perf_counter()
                done = 0
                while time.perf_counter() - t1 < 1.0:
                    orc.run(cfg, host_iq, n_all, n_threads=th)
                    done += n_all
                sweep[th] = done * spe / (time.perf_counter() - t1) / 1e6
        all_rate, all_rates, all_reps, ref = timed(n_all, cores, 1.0)
        n_one = n_all if quick else max(1, n_all // 4)   # ~0.6 s of one thread per pass
        one_rate, one_rates, one_reps, _ = timed(n_one, 1, 0.5)
        sweep[cores] = all_rate
        # the same sample doubles as a parity check of the timed GPU outputs
        g = feats[:n_all].cpu().numpy()
        rel = np.abs(g - ref["features"]) / np.maximum(np.abs(ref["features"]), 1e-30)
        if not args.no_check and (rel.max() > 1e-5 or not np.array_equal(occ_host[:n_all], ref["occupancy"])):
            raise SystemExit(f"bench: GPU results differ from the oracle on the CPU sample (rel {rel.max():.3g})")
        build = "gcc -O2 -march=native -ffp-contract=off, built on this box" if native else "gcc -O2 -ffp-contract=off (portable build)"
        protocol = f"{n_warm} warm-up pass{'es' if n_warm > 1 else ''}, median of {len(all_rates)} passes"
        cpu = {"value": all_rate, "unit": "Msamples/s", "cores": cores, "kind": "port",
               "sample": f"first {n_all} epochs ({n_all * spe * 8 / 2**20:.0f} MiB) of " + ("rank 0's" if world > 1 else "the") + f" GPU batch x {all_reps} per pass, "
                         f"oracle/crn_oracle.c (liquid-dsp-style fp32 radix-2 restated; {build}) on {cores} threads, "
                         f"{protocol} (BASELINE.md §3); {cores} threads = what this process may run at once "
                         f"(host: {hw} hardware threads; {why}); thread_sweep_Msamples_s is for information"
                         + (f"; measured on rank 0 after the timed region, the other {world - 1} ranks idle" if world > 1 else ""),
               "thread_sweep_Msamples_s": {str(k): round(v, 1) for k, v in sorted(sweep.items())},
               "passes": all_rates,
               "one_thread": {"value": one_rate, "unit": "Msamples/s", "cores": 1,
                              "sample": f"first {n_one} epochs x {one_reps} per pass, {protocol}: the "
                                        "reference's own topology (one CE pthread, src/extensible_cognitive_radio.cpp:1761-1808)",
                              "passes": one_rates}}
    if world > 1:
        dist.barrier()   # (gloo's --stage-timeout covers a rank 0 that never arrives)

    # ---- N > 1: what every rank measured and what RCCL itself says the communicator is ------------------
    rccl, per_rank = None, None
    if ex is not None:
        mine = {"rank": rank, "device": local_rank, "kernel_ms_mean": kern_ms_mean, "frac": achieved / HBM_PEAK_GBS, "comm": ex.info(),
                "visible": os.environ.get("HIP_VISIBLE_DEVICES", os.environ.get("ROCR_VISIBLE_DEVICES", "all"))}
        everyone = [mine]
        if world > 1:
            everyone = [None] * world
            dist.all_gather_object(everyone, mine)
        comms = [r["comm"] for r in everyone]
        rccl = {"nranks": comms[0]["nranks"], "nranks_seen_by_every_rank": sorted({c["nranks"] for c in comms}),
                "user_ranks": [c["rank"] for c in comms], "rank0_device": comms[0]["rccl_device"],
                "devices": [c["rccl_device"] for c in comms], "pci_bus_ids": [c["pci_bus_id"] for c in comms],
                "version": comms[0]["rccl_version"], "library": comms[0]["library"],
                "gathers_per_rank": sorted({c["gathers"] for c in comms}), "bytes_per_rank_per_gather": comms[0]["bytes_per_rank"],
                "source": "crn_comm_info on every rank: ncclCommCount / ncclCommUserRank / ncclCommCuDevice / ncclGetVersion of the "
                          "communicator the gathers ran on (version 0 = a stand-in library behind $CRN_RCCL_LIB); pci_bus_ids = "
                          "hipDeviceGetPCIBusId of the device RCCL bound each rank to"}
        if rccl["nranks_seen_by_every_rank"] != [world] or rccl["user_ranks"] != list(range(world)):
            raise SystemExit(f"bench: the communicator is not the {world} ranks of this job: {rccl}")
        # N ranks on N different GPUs: real RCCL refuses two ranks per device, and the line must show it did not have to — every rank names
        # another physical device (a stand-in library, version 0, shares one GPU between the ranks by design: tests only)
        # (a rank whose hipDeviceGetPCIBusId failed reports "": unknown, not "the same device" — ADVICE r05.  It then vouches for its
        # device by what RCCL bound it to plus what the launcher let it see; real RCCL has itself refused two ranks on one device.)
        where = [b or f"unknown-pci:rccl_device={dv}:visible={vis}" for b, dv, vis in zip(rccl["pci_bus_ids"], rccl["devices"], [r["visible"] for r in everyone])]
        if "" in rccl["pci_bus_ids"]:
            rccl["pci_bus_ids_note"] = "hipDeviceGetPCIBusId failed on some rank(s): those are told apart by (RCCL device ordinal, visible-device list)"
        if rccl["version"] != 0 and len(set(where)) != world:
            raise SystemExit(f"bench: {world} ranks but {len(set(where))} distinct GPUs: {where}")
        slow = max(everyone, key=lambda r: r["kernel_ms_mean"])
        fast = min(everyone, key=lambda r: r["kernel_ms_mean"])
        per_rank = {"kernel_ms_mean": [r["kernel_ms_mean"] for r in everyone], "kernel_ms_mean_min": fast["kernel_ms_mean"],
                    "kernel_ms_mean_max": slow["kernel_ms_mean"], "slowest_rank": slow["rank"], "frac_slowest_rank": slow["frac"],
                    "frac_fastest_rank": fast["frac"]}

    if rank == 0:
        if os.environ.get("CRN_BENCH_DUMP"):
            print("kernel_ms per step:", " ".join(f"{x:.3f}" for x in kern_ms), file=sys.stderr)
        line = {
            "metric": f"Msamples/s IQ through FFT+energy-detect, {label}; % HBM roofline",
            "value": value, "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": workload, **({"workload_note": workload_note} if workload_note else {}), "fft_len": N, "frames_per_epoch": K, "epochs_per_gpu": E,
                       "epochs_per_step_all_gpus": E * world, "bytes_per_gpu_per_step": algo_bytes,
                       # the headline's three conditions as fields (the workload text is long and gets cut): the batch is 4.4x SURVEY §8(d)
                       # cfgH's 2 GiB; the kernel keeps n of 16 pass-3 rows (the reference channel plan; null = no pruning); and what the
                       # same kernel reaches on cfgH's batch as worded (= alt.cfgH_2GiB_batch.frac; null when that leg did not run)
                       "batch_GiB": round(algo_bytes / 2 ** 30, 4), "pruned_rows": f"{n_kept}/16" if pruned else None,
                       "cfgH_as_worded_frac": alt["cfgH_2GiB_batch"]["frac"] if alt else None,
                       "kernel": info["name"],
                       "parallelism": f"stream-sharded x{world}" + (", " + ex_kind if multi else ""),
                       "streams_per_gpu": len(tstreams),
                       **({"rccl": rccl} if rccl else {}),
                       **({"noise_floor": noise_floor} if noise_floor else {}),
                       "alt": alt},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                         "kernel_ms_mean": kern_ms_mean, "kernel_ms_min": float(np.min(kern_ms)),
                         "kernel_ms_median": float(np.median(kern_ms)), "kernel_ms_max": float(np.max(kern_ms)),
                         "events": ("span over two streams (launches overlap: first launch to last event) / steps" if two_streams
                                    else "span" if args.span_events else "per-launch"),
                         **({"per_rank": per_rank} if per_rank else {})},
            "cpu_baseline": cpu,
        }
        if roofline_valu is not None:
            line["roofline_valu"] = roofline_valu
            line["roofline"]["note"] = "this mode is VALU-issue-bound, not HBM-bound: see roofline_valu"
        print(json.dumps(line), file=json_out, flush=True)
    stage("shutdown")
    if ex is not None:
        torch.cuda.synchronize()
        ex.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if dog is not None:
        dog.stop()


if __name__ == "__main__":
    try:
        main()
    except SystemExit as e:
        if e.code not in (0, None):
            fail_report(e.code if isinstance(e.code, str) else f"exit {e.code}: {getattr(e, 'crn_reason', 'see the message above')}")
        raise
    except BaseException as e:   # noqa: BLE001 — reported, then re-raised unchanged
        fail_report(f"{type(e).__name__}: {e}")
        raise
