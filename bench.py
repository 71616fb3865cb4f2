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

Multi-GPU: the path shards by stream — every rank owns its own epochs (weak
scaling, per-GPU work fixed) and the only exchange is an RCCL all-gather of the per-epoch
occupancy vector after each step.

Warm-up: W untimed steps, topped up to ~50 ms of launches when W is small (clock ramp), then
exactly K timed steps between barrier + synchronize pairs.

Prints ONE JSON line on rank 0 (contract in the task statement), including
  roofline     HBM-bound: algorithmic bytes (8 B per input sample) / mean kernel time, measured with
               events on the launch stream, against the 8 TB/s peak
  cpu_baseline the oracle (CPU restatement of the reference path) timed on this box's host cores
               on a bounded sample of the same data (rank 0, N = 1 only)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "cognitive-radio-network_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--fft", type=int, default=4096, help="FFT length N (headline: 4096)")
    ap.add_argument("--epochs", type=int, default=0, help="decision epochs per GPU per step (0 = 8.75 GiB of IQ)")
    ap.add_argument("--mode", choices=["energy", "ref", "welch", "scan"], default="energy",
                    help="scan = cfg4's wideband scan: the welch kernel, streams of 64 channels sharded over the ranks")
    ap.add_argument("--variant", type=int, default=0, help="kernel variant (0 = default)")
    ap.add_argument("--frames", type=int, default=0, help="frames per epoch K (0 = the configuration's own, 10)")
    ap.add_argument("--cpu-epochs", type=int, default=-1, help="oracle sample size (-1 = auto, 0 = skip)")
    ap.add_argument("--per-launch-events", action="store_true", help="time each launch with its own event pair")
    ap.add_argument("--no-check", action="store_true", help="skip output checks (ablation variants only)")
    ap.add_argument("--force-collective", action="store_true",
                    help="dry run of the N>1 code path on one GPU: RCCL group of one rank, all-gather every step")
    ap.add_argument("--traffic-json", default=os.path.join(ROOT, "profiles", "hbm_traffic.json"))
    args = ap.parse_args()

    # stdout carries exactly one JSON line: everything else this process (or a library under it:
    # RCCL prints a version banner at init) writes to fd 1 goes to stderr instead.
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    import numpy as np
    import torch
    import torch.distributed as dist

    import crnsense as cs
    from sharding import OccupancyExchange, shard

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: libcrnsense has no CPU path")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    multi = world > 1 or args.force_collective
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=dev)

    if args.mode == "ref":
        cfg = cs.cfg_reference()
        workload = "cfg3: 512-pt |X| mean x10 + 4-5-3 ANN cascade (reference-exact)"
    elif args.mode in ("welch", "scan"):
        cfg = cs.cfg_welch(args.fft, 8, 64)
        for b in range(64):
            cfg.thresh[b] = 1e-2
        workload = f"cfg2: {args.fft}-pt Welch PSD (Hann, 50% overlap) x 64 bands + threshold"
        if args.mode == "scan":
            # BASELINE.json configs[4]: 256 channels = 4 streams x 64 bands is the shard unit; a rank's batch
            # is thousands of such units (one epoch of one stream each), the occupancy gather is [epochs, 64]
            workload = (f"cfg4: wideband scan, {args.fft}-pt Welch x 64 channels per stream (256 channels = 4 streams), "
                        "streams sharded over the ranks, occupancy all-gather")
    else:
        cfg = cs.cfg_energy_scaled(args.fft, 4.0)
        workload = f"{args.fft}-pt FFT + energy detect x 3ch (+noise-floor band), K=10, threshold"
    cfg.device = local_rank
    if args.frames > 0:
        cfg.frames_per_epoch = args.frames
        workload += f" [K={args.frames}]"
    N, K = cfg.fft_len, cfg.frames_per_epoch
    spe = cs.samples_per_epoch(cfg)
    E = args.epochs if args.epochs > 0 else (28672 * 40960) // spe   # per GPU (weak scaling)
    n_samples = cs.samples_needed(cfg, E)
    lo, hi = shard(E * world, rank, world)
    assert hi - lo == E

    sensor = cs.Sensor(cfg)
    sensor.set_variant(args.variant)
    info = sensor.kernel_info()

    iq = torch.zeros(n_samples * 2, dtype=torch.float32, device=dev)
    truth = torch.empty(E, dtype=torch.int32, device=dev)
    feats = torch.empty(E, cfg.n_bands, dtype=torch.float32, device=dev)
    occ = torch.empty(E, cfg.n_bands, dtype=torch.uint8, device=dev)
    dec = torch.empty(E, dtype=torch.int32, device=dev)
    ann = torch.empty(E, 3, dtype=torch.float64, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    sensor.synth_fill_device(iq.data_ptr(), E, spe, seed=0xC0FFEE + 1000 * rank, truth_ptr=truth.data_ptr(),
                             stream=stream)
    outs = {"features": feats.data_ptr(), "ann_out": ann.data_ptr(), "decision": dec.data_ptr(),
            "occupancy": occ.data_ptr(), "spectrum": 0}
    # N > 1: the occupancy block alternates between two slots so that the all-gather of step i
    # (side stream) overlaps the sensing kernel of step i + 1 (sharding.OccupancyExchange)
    ex = OccupancyExchange(E, cfg.n_bands, dev) if multi else None
    n_done = 0

    def step():
        nonlocal n_done
        if multi:
            outs["occupancy"] = ex.local(n_done).data_ptr()
        sensor.run_device(iq.data_ptr(), E, N, outs, stream=stream)
        if multi:
            ex.exchange(n_done)
        n_done += 1

    # Clock ramp: on this part a cold process needs ~25 ms of back-to-back launches before the
    # per-launch time settles, so the W warm-up steps are topped up to at least
    # ~50 ms of untimed work when W is small.  The timed region below is exactly K steps.
    prewarm = max(args.warmup, int(0.05 / 1.6e-3 * (28672 * 40960) / max(E * spe, 1)) + 1)
    for _ in range(prewarm):
        step()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    span = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))

    torch.cuda.synchronize()
    if multi:
        dist.barrier()
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    span[0].record()
    for i in range(args.steps):
        if multi:
            outs["occupancy"] = ex.local(n_done).data_ptr()
        if multi or args.per_launch_events:
            ev[i][0].record()
        sensor.run_device(iq.data_ptr(), E, N, outs, stream=stream)
        if multi or args.per_launch_events:
            ev[i][1].record()
        if multi:
            ex.exchange(n_done)
        n_done += 1
    if multi:
        ex.finish()
        occ = ex.local_bufs[(n_done - 1) % ex.depth]
    span[1].record()
    torch.cuda.synchronize()
    if multi:
        dist.barrier()
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if multi:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    # Kernel duration from events on the launch stream.  One GPU: one pair around the K back-to-back
    # launches (mean includes the ~us dispatch gap between consecutive kernels, no per-event cost).
    # Several GPUs: a pair per launch, because the all-gather sits between launches.
    if multi or args.per_launch_events:
        kern_ms = [a.elapsed_time(b) for a, b in ev]
    else:
        kern_ms = [span[0].elapsed_time(span[1]) / args.steps]
    kern_ms_mean = float(np.mean(kern_ms))
    samples_per_step = E * spe * world
    value = samples_per_step * args.steps / dt / 1e6  # Msamples/s, whole job
    algo_bytes = E * spe * 8                          # per launch: 8 B per unique input sample
    achieved = algo_bytes / (kern_ms_mean * 1e-3) / 1e9

    # sanity on the timed outputs: decisions must follow the driven occupancy pattern
    picked = truth.cpu().numpy()
    if args.no_check:
        pass
    elif cfg.decide == cs.DECIDE_THRESHOLD and cfg.ref_band >= 0:
        o = occ.cpu().numpy()
        want = np.zeros_like(o)
        idx = np.nonzero(picked > 0)[0]
        want[idx, picked[idx]] = 1
        mism = int((o != want).any(axis=1).sum())
        if mism:
            raise SystemExit(f"bench: {mism} epochs whose occupancy differs from the driven pattern")
    elif cfg.decide == cs.DECIDE_ANN:
        mism = int((dec.cpu().numpy() != picked).sum())
        if mism:
            raise SystemExit(f"bench: {mism} epochs whose decision differs from the driven pattern")

    traffic = None
    if os.path.exists(args.traffic_json):
        try:
            tj = json.load(open(args.traffic_json))
            key = f"{'welch' if args.mode == 'scan' else args.mode}{N}"
            if key in tj:  # measured once per kernel; scales linearly with the batch
                traffic = int(tj[key]["hbm_bytes_per_launch"] * (E / tj[key]["epochs"]))
        except Exception:
            traffic = None

    cpu = None
    if rank == 0 and world == 1 and args.cpu_epochs != 0:
        import oracle_py as orc
        cores = os.cpu_count() or 1
        cap = max(1, (7168 * 40960) // spe)   # at most 2.2 GiB of the batch goes to the host
        n_cpu = args.cpu_epochs if args.cpu_epochs > 0 else min(E, cap, 64 * cores)
        host_iq = iq[: cs.samples_needed(cfg, n_cpu) * 2].cpu().numpy()
        orc.run(cfg, host_iq, min(n_cpu, 2 * cores), n_threads=cores)  # warm-up (page in, plan)
        t1 = time.perf_counter()
        ref = orc.run(cfg, host_iq, n_cpu, n_threads=cores)
        t_cpu = time.perf_counter() - t1
        if args.cpu_epochs < 0 and t_cpu < 5.0:  # grow the sample to ~10 s of CPU work
            n_cpu = int(min(E, cap, n_cpu * 10.0 / max(t_cpu, 1e-3)))
            host_iq = iq[: cs.samples_needed(cfg, n_cpu) * 2].cpu().numpy()
            t1 = time.perf_counter()
            ref = orc.run(cfg, host_iq, n_cpu, n_threads=cores)
            t_cpu = time.perf_counter() - t1
        # the same sample doubles as a parity check of the timed GPU outputs
        g = feats[:n_cpu].cpu().numpy()
        rel = np.abs(g - ref["features"]) / np.maximum(np.abs(ref["features"]), 1e-30)
        if rel.max() > 1e-5 or not np.array_equal(occ[:n_cpu].cpu().numpy(), ref["occupancy"]):
            raise SystemExit(f"bench: GPU results differ from the oracle on the CPU sample (rel {rel.max():.3g})")
        cpu = {"value": n_cpu * spe / t_cpu / 1e6, "unit": "Msamples/s", "cores": cores, "kind": "port",
               "sample": f"first {n_cpu} epochs ({n_cpu * spe * 8 / 2**20:.0f} MiB) of the GPU batch, "
                         f"oracle/crn_oracle.c (liquid-dsp-style fp32 radix-2 restated) on {cores} threads, "
                         f"{t_cpu:.1f} s"}

    if rank == 0:
        if os.environ.get("CRN_BENCH_DUMP"):
            print("kernel_ms per step:", " ".join(f"{x:.3f}" for x in kern_ms), file=sys.stderr)
        line = {
            "metric": "Msamples/s IQ through FFT+energy-detect, 4096-pt x 3ch; % HBM roofline",
            "value": value, "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": workload, "fft_len": N, "frames_per_epoch": K, "epochs_per_gpu": E,
                       "bytes_per_gpu_per_step": algo_bytes, "kernel": info["name"],
                       "parallelism": f"stream-sharded x{world}" + (", RCCL all-gather of occupancy" if multi else "")},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel_ms_mean": kern_ms_mean, "kernel_ms_min": float(np.min(kern_ms)),
                         "kernel_ms_median": float(np.median(kern_ms))},
            "cpu_baseline": cpu,
        }
        print(json.dumps(line), file=json_out, flush=True)
    if multi:
        # every rank must find its own block, unchanged, at its place in the gathered vector
        if not args.no_check and not torch.equal(ex.gathered(n_done - 1)[rank * E:(rank + 1) * E], occ):
            raise SystemExit(f"bench: rank {rank}: gathered occupancy differs from the local block")
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
