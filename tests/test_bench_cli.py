"""bench.py's contract (one JSON line on stdout with roofline and cpu_baseline), and the N > 1 code path
dry-run on one GPU: RCCL group of one rank, occupancy all-gather overlapped with the next launch."""
import json
import os
import subprocess
import sys

import pytest

sys.path[:0] = [os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cognitive-radio-network_amd")]
import crnsense as cs  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _run(*extra, epochs=("--epochs", "512")):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541")
    from test_comm import _run_group   # own process group, one retry after a timeout (a communicator bring-up hung once on one box)
    out = _run_group([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "2",
                      *epochs, *(() if "--live" in extra else ("--no-live-traffic",)),
                      *[e for e in extra if e != "--live"]], 300, env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, out.stdout            # exactly one line, whatever the libraries print
    return json.loads(lines[0])


def test_bench_prints_one_json_line_with_roofline_and_cpu_baseline(built):
    d = _run("--cpu-epochs", "64")
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["scaling"] == "weak" and d["vs_baseline"] is None
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and 0.0 < r["frac"] < 1.0
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and "epochs" in c["sample"]
    assert c["one_thread"]["cores"] == 1 and c["one_thread"]["value"] > 0
    assert r["events"] == "per-launch" and r["kernel_ms_min"] <= r["kernel_ms_median"] <= r["kernel_ms_max"]
    assert d["metric"].startswith("Msamples/s IQ through FFT+energy-detect, 4096-pt x 3ch")
    assert d["config"]["workload"].startswith("4096-pt")
    # the driver's record keeps 120 characters of it: the disclosure that matters (row pruning) must be inside them; the long form is a field
    assert len(d["config"]["workload"]) <= 120 and "pruned: 7/16 pass-3 rows" in d["config"]["workload"]
    assert "specialised to the reference channel plan" in d["config"]["workload_note"]
    # the headline's conditions as fields: batch size, pass-3 row pruning, and — when the 2 GiB leg ran — cfgH as worded
    c = d["config"]
    assert c["batch_GiB"] == round(512 * 40960 * 8 / 2 ** 30, 4) and c["pruned_rows"] == "7/16" and c["cfgH_as_worded_frac"] is None
    assert d["roofline"]["kernel_ms_mean"] <= d["ms_per_step"]


def test_bench_default_batch_with_every_alt_leg_and_the_cpu_check(built):
    """The driver's shape: the default 8.75 GiB batch, so that the config.alt legs run (SURVEY §8(d)'s 2 GiB batch on one and on two
    streams, the unpruned kernel) — and the CPU sample check after them still reads the headline launch's results, not a leg's.
    Nothing outside north_star rides in the default line: no 16-bit-input or wire-format legs (--wire-format is opt-in)."""
    d = _run("--cpu-epochs", "64", epochs=())
    alt = d["config"]["alt"]
    assert set(alt) == {"cfgH_2GiB_batch", "cfgH_2GiB_batch_two_streams", "unpruned"}
    two = alt.pop("cfgH_2GiB_batch_two_streams")
    # (north_star's 70 % on every leg even on this short run; the tight floors, on the driver's shape and with re-measurement, are
    # tests/test_zz_roofline_floors.py)
    import perf_floors as pf
    assert two["outputs_identical_across_streams"] and two["launches"] == 60 and pf.NORTH_STAR < two["frac"] < 1.0
    assert two["bytes_per_step"] == alt["cfgH_2GiB_batch"]["bytes_per_step"]
    assert alt["cfgH_2GiB_batch"]["kernel_ms_min"] <= alt["cfgH_2GiB_batch"]["kernel_ms_mean"]
    for leg in alt.values():
        assert pf.NORTH_STAR < leg["frac"] < 1.0 and leg["kernel_ms_mean"] > 0
    assert pf.NORTH_STAR < d["roofline"]["frac"] < 1.0 and d["roofline"]["kernel_ms_mean"] <= d["ms_per_step"]
    assert alt["cfgH_2GiB_batch"]["bytes_per_step"] == 6553 * 40960 * 8
    assert d["config"]["epochs_per_gpu"] == 28672 and d["cpu_baseline"]["value"] > 0
    assert d["config"]["batch_GiB"] == 8.75 and d["config"]["pruned_rows"] == "7/16"
    assert d["config"]["cfgH_as_worded_frac"] == alt["cfgH_2GiB_batch"]["frac"]
    assert list(d["config"]).index("cfgH_as_worded_frac") < list(d["config"]).index("alt")      # ahead of the long fields: survives a cut-off tail


def test_bench_measures_hbm_traffic_in_the_run(built):
    """roofline.traffic comes from rocprofv3 counter passes of a short child run of the same workload (FETCH_SIZE x 2 +
    WRITE_SIZE), not from a committed file: within 1 % of the algorithmic bytes for the read-once kernel."""
    import shutil
    if shutil.which("rocprofv3") is None:
        pytest.skip("rocprofv3 not on PATH")
    d = _run("--cpu-epochs", "0", "--live")
    r = d["roofline"]
    assert r["traffic_source"].startswith("measured in this run"), r["traffic_source"]
    assert abs(r["traffic"] / d["config"]["bytes_per_gpu_per_step"] - 1) < 0.01, r


@pytest.mark.parametrize("mode", [[], ["--mode", "ref"], ["--mode", "welch"], ["--mode", "scan"]])
def test_bench_collective_path_on_one_gpu(built, mode):
    d = _run("--cpu-epochs", "0", "--force-collective", *mode)
    assert "RCCL all-gather of occupancy" in d["config"]["parallelism"]
    assert d["value"] > 0 and d["cpu_baseline"] is None
    # one real RCCL rank: the line says what RCCL itself reports for the communicator the gathers ran on
    r = d["config"]["rccl"]
    assert r["nranks"] == 1 and r["user_ranks"] == [0] and r["version"] > 20000 and "rccl" in r["library"] and r["rank0_device"] == 0
    assert r["gathers_per_rank"][0] >= d["steps"] and d["roofline"]["per_rank"]["slowest_rank"] == 0
    if "welch" in mode or "scan" in mode:
        assert d["roofline_valu"]["bound"] == "valu" and 0 < d["roofline_valu"]["frac"] < 1
        assert "Welch" in d["metric"]


def test_bench_two_streams_per_gpu(built):
    """--two-streams (what --scaling strong selects by itself for shares under 4 GiB): launches alternate between two streams with an
    output set each and four exchange slots; the roofline is priced from the span over both streams; the last launch's outputs still
    pass the oracle check and the gathered-vector check."""
    d = _run("--cpu-epochs", "0", "--two-streams", "--force-collective")
    assert d["config"]["streams_per_gpu"] == 2 and "two streams" in d["roofline"]["events"]
    assert d["config"]["rccl"]["nranks"] == 1 and d["config"]["rccl"]["gathers_per_rank"][0] >= d["steps"]
    assert 0 < d["roofline"]["frac"] < 1 and d["roofline"]["kernel_ms_min"] == d["roofline"]["kernel_ms_max"]   # one span figure
    d = _run("--cpu-epochs", "0", "--two-streams", "--mode", "welch")
    assert d["config"]["streams_per_gpu"] == 2 and d["value"] > 0


def test_bench_ranks_stop_together_when_rccl_cannot_be_loaded(built):
    """The occupancy exchange has ONE backend — RCCL through the C ABI's crn_comm_*.  A job whose ranks cannot load it ($CRN_RCCL_LIB
    names a file that is not there) agrees on that over the control group and every rank exits non-zero: no hang inside
    ncclCommInitRank, no other collective backend, no JSON line."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(CRN_RCCL_LIB="/nonexistent/librccl.so", HIP_VISIBLE_DEVICES="0")
    from test_comm import _run_group
    out = _run_group([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--epochs", "256",
                      "--cpu-epochs", "0"], 300, env, cwd=ROOT)
    assert out.returncode != 0 and not [ln for ln in out.stdout.splitlines() if ln.strip().startswith("{")]
    assert "RCCL is not usable through crn_comm_*" in out.stderr or "cannot load RCCL" in out.stderr
    from test_bench_launch import FAILURE_KEYS, failure_reports
    reps = failure_reports(out.stderr)          # ... and each rank that got to say so names the stage and what it could see
    assert reps and all(set(f) == FAILURE_KEYS and f["failed_stage"].startswith("communicator") and "RCCL" in f["why"] and f["rccl"] is None
                        and f["visible_devices"]["HIP_VISIBLE_DEVICES"] == "0" and f["pci_bus_id"] for f in reps), out.stderr[-2000:]
    import sharding
    assert not hasattr(sharding, "TorchOccupancyExchange")


def test_bench_wire_format_mode(built):
    """--wire-format: samples held as int16 pairs; the oracle check of the run still applies (it compares against the float samples the
    wire buffer was packed from), bytes per step are 4 per sample, and the workload says it is not the headline configuration.
    (bench.py loads the optional libcrnsense_sc16.so for it.)"""
    if not os.path.exists(os.path.join(ROOT, "cognitive-radio-network_amd", "libcrnsense_sc16.so")):
        pytest.skip("libcrnsense_sc16.so was not built (make -C csrc SC16=1)")
    d = _run("--cpu-epochs", "0", "--wire-format")
    assert d["config"]["bytes_per_gpu_per_step"] == 512 * 40960 * 4
    assert "NOT THE HEADLINE CONFIGURATION" in d["config"]["workload"] and d["roofline"]["traffic"] is None   # (--no-live-traffic here)
    d = _run("--cpu-epochs", "0", "--wire-format", "--live")                  # measured in the run: 4 B per sample, read once
    assert d["roofline"]["traffic_source"].startswith("measured in this run")
    assert 1.0 <= d["roofline"]["traffic"] / d["config"]["bytes_per_gpu_per_step"] < 1.02
    d = _run("--cpu-epochs", "0", "--wire-format", "--mode", "welch")
    assert "wire format" in d["config"]["workload"]


def test_bench_two_ranks_on_one_gpu_through_a_stand_in_rccl(built):
    """bench.py's whole N > 1 flow on hardware, the way the driver launches it (`python -m torch.distributed.run --nproc-per-node 2 ..
    bench.py --gpus 2`): gloo control plane, unique id made by rank 0 through the C ABI and broadcast, streams sharded over the
    ranks, crn_comm_* slots, barriers, max-over-ranks timing, every rank finding its block at its place in the gathered vector, one
    JSON line from rank 0.  Real RCCL refuses two ranks on one device and the pool hands out one GPU: the wire is
    tests/harness/libfake_rccl_mp.so (shared memory between the two processes), everything above it is the product."""
    fake = os.path.join(ROOT, "tests", "harness", "libfake_rccl_mp.so")
    assert os.path.exists(fake)
    env = dict(os.environ, CRN_RCCL_LIB=fake, HIP_VISIBLE_DEVICES="0")
    for extra in ([], ["--mode", "scan"]):
        from test_comm import _run_group   # own process group: a timeout must not leave the two ranks behind
        out = _run_group([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", "29577", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2",
                          "--epochs", "2048", "--cpu-epochs", "32", *extra], 300, env, cwd=ROOT)
        assert out.returncode == 0, _why(out.stderr)
        lines = [ln for ln in out.stdout.splitlines() if ln.strip().startswith("{")]
        assert len(lines) == 1, out.stdout[-2000:]
        d = json.loads(lines[0])
        assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["epochs_per_gpu"] == 2048
        assert d["config"]["parallelism"].startswith("stream-sharded x2") and "crn_comm_" in d["config"]["parallelism"]
        assert d["value"] > 0 and d["ms_per_step"] > 0
        _check_cpu_baseline_of_an_n_gt_1_line(d, 2)
        assert d["config"]["rccl"]["nranks"] == 2 and d["config"]["rccl"]["user_ranks"] == [0, 1]
        assert len(d["config"]["rccl"]["pci_bus_ids"]) == 2


def _check_cpu_baseline_of_an_n_gt_1_line(d, n):
    """north_star: throughput at 1 / 2 / 4 / 8 GPUs NEXT TO the CPU path timed on the same box's host cores (count stated)."""
    c = d["cpu_baseline"]
    assert c is not None and c["value"] > 0 and c["cores"] >= 1 and c["kind"] == "port" and c["unit"] == "Msamples/s"
    assert c["one_thread"]["value"] > 0 and c["one_thread"]["cores"] == 1
    assert "rank 0" in c["sample"] and f"the other {n - 1} ranks idle" in c["sample"] and "threads" in c["sample"]
    assert "not re-measured at N > 1" in (d["roofline"]["traffic_source"] or "not re-measured at N > 1")


def _why(stderr):
    """What a failed N > 1 run said: the ranks' own failure objects and watchdog lines first (they name the stage), then the tail."""
    from test_bench_launch import failure_reports
    said = [ln for ln in stderr.splitlines() if "did not finish within" in ln or ln.startswith("sharding:") or ln.startswith("bench")]
    return json.dumps(failure_reports(stderr)) + "\n" + "\n".join(said[-16:]) + "\n" + stderr[-3000:]


def _self_launched(n, *extra, epochs="512", timeout=600, cpu_epochs="0"):
    """`python bench.py --gpus n` with NO launcher and no WORLD_SIZE in the environment — the shape of the driver's single-GPU command
    at n > 1 — on the one GPU of the box, the ranks over tests/harness/libfake_rccl_mp.so."""
    fake = os.path.join(ROOT, "tests", "harness", "libfake_rccl_mp.so")
    assert os.path.exists(fake)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(CRN_RCCL_LIB=fake, HIP_VISIBLE_DEVICES="0")
    from test_comm import _run_group
    # (--stage-timeout 90: these runs take 5 - 30 s in all; a rank that hangs must cost the suite a minute and a half, not five)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "4", "--warmup", "2",
           "--epochs", epochs, "--cpu-epochs", cpu_epochs, "--stage-timeout", "90", *extra]
    out = _run_group(cmd, timeout, env, cwd=ROOT)
    from test_bench_launch import failure_reports
    if out.returncode != 0 and any("watchdog" in f["why"] for f in failure_reports(out.stderr)):
        # _run_group's policy for a run that hangs until ITS timeout (one more try: a communicator bring-up that hung on a bad box has
        # been seen once) extended to a run the ranks' own watchdog ended first — said out loud, so that the log shows it happened
        print("FIRST ATTEMPT ENDED BY THE STAGE WATCHDOG, trying once more:\n" + _why(out.stderr)[:4000])
        out = _run_group(cmd, timeout, env, cwd=ROOT)
    assert out.returncode == 0, _why(out.stderr)
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, out.stdout[-2000:]          # the parent relays exactly rank 0's line
    return json.loads(lines[0])


def test_bench_self_launches_two_ranks_without_a_launcher(built):
    d = _self_launched(2)
    assert d["n_gpus"] == 2 and d["config"]["parallelism"].startswith("stream-sharded x2")
    assert "crn_comm_" in d["config"]["parallelism"] and d["config"]["rccl"]["nranks"] == 2


@pytest.mark.gpu_first
@pytest.mark.parametrize("scaling", ["weak", "strong"])
@pytest.mark.parametrize("mode", [[], ["--mode", "scan"]])
def test_bench_cfg4_rehearsal_eight_self_launched_ranks_on_one_gpu(built, mode, scaling):
    """BASELINE.json configs[4] as the driver will start it (`python3 bench.py --gpus 8 ...`), rehearsed with the eight ranks sharing
    the one GPU: bench.py starts torch.distributed.run itself before it touches the GPU, gloo control plane, RCCL unique id from
    rank 0 through the C ABI, streams sharded eight ways, crn_comm_* slots, the barriers and the max-over-ranks timing; EVERY rank
    checks that its own block sits unchanged at its place in the gathered vector (a mismatch on any rank makes the launcher, and so
    the parent, exit non-zero) and rank 0's single JSON line is relayed."""
    # weak: 4096 epochs on every rank.  strong: 32 768 epochs in all, split eight ways — a share far below 4 GiB, so every rank
    # alternates its launches between two streams (four exchange slots).  (Eight rank processes are all the GPU schedules at once:
    # the test is marked gpu_first so that this pytest process is not a ninth — tests/conftest.py — and the batch is large enough
    # that the ~50 ms clock-ramp top-up is ~200 lock-step gathers, not ~900, should a ninth process be there all the same.)
    d = _self_launched(8, *mode, "--scaling", scaling, epochs="4096" if scaling == "weak" else "32768", cpu_epochs="32")
    assert d["n_gpus"] == 8 and d["scaling"] == scaling and d["config"]["epochs_per_gpu"] == 4096
    assert d["config"]["epochs_per_step_all_gpus"] == 32768 and d["config"]["streams_per_gpu"] == (2 if scaling == "strong" else 1)
    assert d["config"]["parallelism"].startswith("stream-sharded x8") and "crn_comm_" in d["config"]["parallelism"]
    assert d["value"] > 0 and d["ms_per_step"] > 0 and d["steps"] == 4
    _check_cpu_baseline_of_an_n_gt_1_line(d, 8)       # every N > 1 line carries the CPU figure (rank 0 measures it after the timed region)
    # the line proves what the collective saw: eight ranks, every one of them counted eight, in rank order, each with its gathers,
    # and which physical GPU each was bound to (here one GPU eight times: the stand-in, version 0, is the only library allowed that)
    r = d["config"]["rccl"]
    assert r["nranks"] == 8 and r["nranks_seen_by_every_rank"] == [8] and r["user_ranks"] == list(range(8)) and len(r["devices"]) == 8
    assert len(r["pci_bus_ids"]) == 8 and all(isinstance(b, str) and len(b) >= 7 and b.count(":") == 2 for b in r["pci_bus_ids"])
    assert len(set(r["pci_bus_ids"])) == 1
    assert "fake_rccl_mp" in r["library"] and r["version"] == 0          # (the stand-in says so)
    pr = d["roofline"]["per_rank"]
    assert len(pr["kernel_ms_mean"]) == 8 and pr["kernel_ms_mean_min"] <= pr["kernel_ms_mean_max"] and 0 <= pr["slowest_rank"] < 8
    assert 0 < pr["frac_slowest_rank"] <= pr["frac_fastest_rank"]
    assert ("two streams" in d["roofline"]["events"]) == (scaling == "strong")
    if mode:
        assert "cfg4" in d["config"]["workload"]


def test_bench_self_launch_propagates_a_rank_failure(built):
    """A rank that dies (here: an argument only the ranks reject) must surface as a non-zero exit of the parent, not as a hang or a
    clean exit without a line."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    from test_comm import _run_group
    out = _run_group([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--epochs", "64",
                      "--cpu-epochs", "0", "--fft", "3000"], 300, env, cwd=ROOT)
    assert out.returncode != 0 and not [ln for ln in out.stdout.splitlines() if ln.strip().startswith("{")]


def test_bench_a_hung_rank_ends_the_job_within_the_stage_timeout(built):
    """Fail-fast at N > 1 (VERDICT r04 next #1b): a rank that HANGS — stopped with SIGSTOP as soon as it exists, so it never joins the
    others — must cost --stage-timeout (300 s by default, 20 s here), not gloo's default 30 minutes: the live rank gives up in
    gloo or at its watchdog, exits non-zero, the launcher kills the stopped one, the parent relays the failure and prints no line."""
    import signal
    import time
    import psutil
    fake = os.path.join(ROOT, "tests", "harness", "libfake_rccl_mp.so")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(CRN_RCCL_LIB=fake, HIP_VISIBLE_DEVICES="0")
    t0 = time.monotonic()
    p = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--epochs", "256",
                          "--cpu-epochs", "0", "--stage-timeout", "20"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env,
                         cwd=ROOT, start_new_session=True)
    stopped = None
    try:
        while stopped is None and time.monotonic() - t0 < 120 and p.poll() is None:
            for ch in psutil.Process(p.pid).children(recursive=True):
                try:
                    if ch.environ().get("RANK") == "1" and "bench.py" in " ".join(ch.cmdline()):
                        ch.send_signal(signal.SIGSTOP)
                        stopped = ch.pid
                        break
                except (psutil.NoSuchProcess, psutil.AccessDenied):
                    pass
            time.sleep(0.005)
        assert stopped is not None, "rank 1 never appeared"
        out, err = p.communicate(timeout=150)
    finally:
        try:
            os.killpg(p.pid, signal.SIGKILL)
        except ProcessLookupError:
            pass
    took = time.monotonic() - t0
    assert p.returncode != 0 and not [ln for ln in out.splitlines() if ln.strip().startswith("{")], (p.returncode, out[-500:])
    assert took < 120, took      # stage timeout 20 s + the launcher's own 30 s grace before it kills the stopped rank
    # the live rank's one-object account of where it gave up: the stage, how long in, what it could see
    from test_bench_launch import FAILURE_KEYS, failure_reports
    reps = failure_reports(err)
    assert [f["rank"] for f in reps] == [0], err[-2000:]
    f = reps[0]
    assert set(f) == FAILURE_KEYS and f["world"] == 2 and f["failed_stage"] and f["elapsed_s"] >= 15 and f["why"]
    assert f["visible_devices"]["HIP_VISIBLE_DEVICES"] == "0" and f["visible_devices"]["device_count"] == 1 and f["pci_bus_id"]
    assert not psutil.pid_exists(stopped) or psutil.Process(stopped).status() == psutil.STATUS_ZOMBIE


def test_bench_two_real_rccl_ranks_on_one_gpu_are_refused_not_hung(built):
    """The furthest real RCCL goes on a one-GPU box: two self-launched ranks, the unique id made by rank 0 through the C ABI and broadcast
    over gloo, both ranks inside ncclCommInitRank talking to each other over RCCL's own bootstrap — where RCCL finds the same PCI device
    twice and refuses ("invalid usage").  The failure must come back as a message naming the call and a non-zero exit of the parent
    within seconds, no JSON line, no rank left behind: the N > 1 path fails fast with the real library too."""
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "CRN_RCCL_LIB")}
    env.update(HIP_VISIBLE_DEVICES="0")
    from test_comm import _run_group
    t0 = time.monotonic()
    out = _run_group([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--epochs", "256",
                      "--cpu-epochs", "0", "--stage-timeout", "40"], 200, env, cwd=ROOT)
    took = time.monotonic() - t0
    assert out.returncode != 0 and not [ln for ln in out.stdout.splitlines() if ln.strip().startswith("{")], out.stdout[-500:]
    assert "ncclCommInitRank" in out.stderr, out.stderr[-2000:]
    assert took < 120, took
    # each rank RCCL refused says so in one object: the stage, the call that failed, the one device both ranks were shown (the cause)
    from test_bench_launch import FAILURE_KEYS, failure_reports
    reps = failure_reports(out.stderr)
    assert reps and {f["rank"] for f in reps} <= {0, 1}, out.stderr[-2000:]
    for f in reps:
        assert set(f) == FAILURE_KEYS and f["failed_stage"].startswith("communicator") and "ncclCommInitRank" in f["why"] and f["rccl"] is None
        assert f["visible_devices"]["HIP_VISIBLE_DEVICES"] == "0" and f["visible_devices"]["device_count"] == 1 and f["pci_bus_id"]
    assert len({f["pci_bus_id"] for f in reps}) == 1
    d = os.environ.get("CRN_EVIDENCE_DIR")
    if d and os.path.isdir(d):
        msg = [ln for ln in out.stderr.splitlines() if "ncclCommInitRank" in ln and not ln.lstrip().startswith("{")]
        open(os.path.join(d, "two_real_rccl_ranks_one_gpu.txt"), "w").write(
            f"bench.py --gpus 2 with the real RCCL on one GPU: exit {out.returncode} after {took:.1f} s, no JSON line\n" + "\n".join(msg[:4]) + "\n"
            + "\n".join(json.dumps({"bench_failure": f}) for f in reps) + "\n")
