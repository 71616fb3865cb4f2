"""bench.py's self-launch (no GPU needed): `python bench.py --gpus N` without WORLD_SIZE must start
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py <same args>` as a
CHILD process before importing torch or touching the GPU, relay rank 0's JSON line and the exit code."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load_bench():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def failure_reports(stderr):
    """The `{"bench_failure": {...}}` objects failing ranks wrote to stderr (bench.py: fail_report), one per line — each once: a
    self-launching parent says them again at the end of its own stderr."""
    out, seen = [], set()
    for ln in stderr.splitlines():
        ln = ln.strip()
        if ln.startswith('{"bench_failure"') and ln not in seen:
            seen.add(ln)
            out.append(json.loads(ln)["bench_failure"])
    return out


FAILURE_KEYS = {"failed_stage", "rank", "world", "elapsed_s", "why", "rccl", "visible_devices", "pci_bus_id", "device"}


def test_self_launch_command_and_relay(monkeypatch, capsys):
    bench = _load_bench()
    seen = {}

    def fake_run(cmd, env):
        seen["cmd"], seen["kw"] = cmd, {"env": env}
        return 0, "RCCL banner that a library printed\n" + json.dumps({"n_gpus": 8, "value": 1.0}) + "\n", []

    monkeypatch.setattr(bench, "run_ranks", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "3"])
    assert "torch" not in sys.modules or True     # (pytest plugins may have imported it; bench.py itself must not need it here)
    rc = bench.self_launch(8)
    out = capsys.readouterr()
    assert rc == 0
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "8" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert 1024 <= int(cmd[cmd.index("--master-port") + 1]) < 65536
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "8", "--steps", "3"]          # the ranks get the caller's arguments unchanged
    assert seen["kw"]["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert [ln for ln in out.out.splitlines() if ln.strip()] == [json.dumps({"n_gpus": 8, "value": 1.0})]   # stdout: the one line
    assert "RCCL banner" in out.err


def test_self_launch_exit_codes(monkeypatch, capsys):
    bench = _load_bench()

    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2"])
    monkeypatch.setattr(bench, "run_ranks", lambda cmd, env: (3, "", []))
    assert bench.self_launch(2) == 3
    monkeypatch.setattr(bench, "run_ranks", lambda cmd, env: (0, "no json here\n", []))
    assert bench.self_launch(2) == 1
    capsys.readouterr()


def test_self_launch_repeats_the_ranks_failure_objects_last(monkeypatch, capfd):
    """The launcher's own failure summary (dozens of lines per job) follows the ranks' one-line accounts: the parent passes stderr through
    as it comes — nothing is held back from a caller that kills a hung job — and says the failure objects again at the very end."""
    bench = _load_bench()
    obj = json.dumps({"bench_failure": {"failed_stage": "communicator", "rank": 1, "world": 2, "why": "x"}})
    child = ("import sys, time\n"
             "sys.stderr.write('early line\\n'); sys.stderr.flush()\n"
             f"sys.stderr.write({obj!r} + '\\n')\n"
             "sys.stderr.write('launcher summary line\\n' * 50)\n"
             "print('not json'); sys.exit(5)\n")
    rc, out, failures = bench.run_ranks([sys.executable, "-c", child], dict(os.environ))
    assert rc == 5 and out.strip() == "not json" and failures == [obj]
    err = capfd.readouterr().err
    assert err.startswith("early line") and err.count("launcher summary line") == 50 and obj in err
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2"])
    monkeypatch.setattr(bench, "run_ranks", lambda cmd, env: (5, "not json\n", [obj]))
    assert bench.self_launch(2) == 5
    err = capfd.readouterr().err
    assert err.rstrip().splitlines()[-1] == obj and "what the failing rank(s) said" in err


def test_main_self_launches_before_any_gpu_or_torch_use(monkeypatch):
    """main() with --gpus 4 and no WORLD_SIZE goes to self_launch() and exits with its code; with WORLD_SIZE set (a rank
    started by the launcher) it does not."""
    bench = _load_bench()
    calls = []
    monkeypatch.setattr(bench, "self_launch", lambda n: calls.append(n) or 7)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    try:
        bench.main()
        raise AssertionError("main() returned")
    except SystemExit as e:
        assert e.code == 7
    assert calls == [4]


def test_two_self_launched_ranks_reach_the_gpu_check_without_a_gpu():
    """End to end without a GPU: the parent starts two real rank processes; each stops at "bench.py needs a GPU" (libcrnsense has no
    CPU path) and the parent reports the failure — no hang, no JSON line."""
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("GPU present: covered by tests/test_bench_cli.py")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, timeout=300)
    assert r.returncode != 0 and "{" not in r.stdout
    assert "needs a GPU" in r.stderr
    # every failing rank of an N > 1 job names itself, the stage and what it could see, as one JSON object on stderr
    reps = failure_reports(r.stderr)
    # (the launcher stops the other rank as soon as the first one has failed: one report or two)
    assert reps and {f["rank"] for f in reps} <= {0, 1} and len({f["rank"] for f in reps}) == len(reps), r.stderr[-2000:]
    for f in reps:
        assert set(f) == FAILURE_KEYS and f["world"] == 2 and f["failed_stage"] == "start" and "needs a GPU" in f["why"]
        assert f["rccl"] is None and f["pci_bus_id"] is None and f["elapsed_s"] >= 0
        assert f["visible_devices"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"      # (self_launch exports it for the ranks)
    # ... and the parent says the objects again after the launcher's own summary: the last line of stderr is one of them
    assert r.stderr.rstrip().splitlines()[-1].startswith('{"bench_failure"') and "ChildFailedError" in r.stderr


def test_failure_report_is_silent_at_n_1_and_one_line_otherwise(capfd):
    bench = _load_bench()
    bench.RUN.update(world=1, multi=False)
    bench.fail_report("anything")
    assert capfd.readouterr().err == ""
    bench.RUN.update(world=8, rank=5, multi=True, stage="communicator (ncclCommInitRank is collective)", device=5, devices_seen=8,
                     pci_bus_id="0000:c5:00.0", rccl={"nranks": 8, "rank": 5, "pci_bus_id": "0000:c5:00.0"})
    bench.fail_report("CrnError: ncclCommInitRank: unhandled system error")
    err = capfd.readouterr().err
    assert err.count("\n") == 1
    (f,) = failure_reports(err)
    assert set(f) == FAILURE_KEYS and f["rank"] == 5 and f["world"] == 8 and f["failed_stage"].startswith("communicator")
    assert f["rccl"]["nranks"] == 8 and f["pci_bus_id"] == "0000:c5:00.0" and f["visible_devices"]["device_count"] == 8
    assert "ncclCommInitRank" in f["why"]


def test_watchdog_ends_a_rank_stuck_in_a_stage():
    """bench.py's per-stage watchdog (N > 1): a rank whose stage outlives --stage-timeout says where it was and exits 6 — from a thread,
    because the main thread is blocked (here: asleep; in a real run inside ncclCommInitRank or a device synchronise)."""
    code = ("import importlib.util, time\n"
            f"spec = importlib.util.spec_from_file_location('b', {os.path.join(ROOT, 'bench.py')!r})\n"
            "b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)\n"
            "b.RUN.update(world=4, rank=3)\n"
            "d = b.Watchdog(0.5, 3)\n"
            "d.pet('first'); time.sleep(0.2); d.pet('the stage that hangs'); time.sleep(30)\n")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60)
    assert r.returncode == 6 and "rank 3" in r.stderr and "the stage that hangs" in r.stderr
    (f,) = failure_reports(r.stderr)
    assert f["failed_stage"] == "the stage that hangs" and f["rank"] == 3 and f["world"] == 4 and "watchdog" in f["why"] and f["elapsed_s"] >= 0.5
    code = code.replace("time.sleep(30)", "d.stop(); time.sleep(1.5)")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr


def test_eight_rank_rehearsals_are_collected_ahead_of_everything_that_opens_the_device():
    """tests/conftest.py moves the gpu_first tests (eight rank processes on the one GPU) to the front of ANY selection, so that the pytest
    process — which opens the device in its first in-process GPU test — is never a ninth GPU process beside them
    (profiles/r06_nine_gpu_processes.txt).  Here: files named in the opposite order."""
    r = subprocess.run([sys.executable, "-m", "pytest", "--collect-only", "-q", "-m", "gpu", "-p", "no:cacheprovider",
                        os.path.join(ROOT, "tests", "test_zz_roofline_floors.py"), os.path.join(ROOT, "tests", "test_golden.py"),
                        os.path.join(ROOT, "tests", "test_bench_cli.py")], capture_output=True, text=True, timeout=300, cwd=ROOT)
    ids = [ln for ln in r.stdout.splitlines() if "::" in ln]
    assert r.returncode == 0 and len(ids) > 20, r.stdout[-2000:] + r.stderr[-2000:]
    first = [i for i in ids if "cfg4_rehearsal_eight_self_launched_ranks" in i]
    assert len(first) == 4 and ids[:4] == first, ids[:8]
    assert "test_zz_roofline_floors" in ids[4]           # the rest keeps the order it was given
