"""The occupancy exchange through the C ABI (crn_comm_* in include/crn_sense.h: RCCL all-gather on a side
stream, `depth` slots).  On the GPU box: a one-rank communicator end to end (create, slot hand-out, gather,
slot reuse, ordering against the launch stream) and the sensing kernel writing straight into a slot.  Without a
GPU: the entry points fail cleanly.  The two-rank layout (rank order, slot reuse) is covered on CPU by
tests/test_sharding_gloo.py with the gloo twin of the same slot logic."""
import ctypes as C
import os

import numpy as np
import pytest

import crnsense as cs


def test_comm_fails_cleanly_without_gpu(built):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    try:                      # whether RCCL hands out an id without a device depends on its version:
        uid = cs.comm_unique_id()   # either a 128-byte id ...
        assert len(uid) == cs.COMM_ID_BYTES
    except cs.CrnError:       # ... or a clean error; creating a communicator must fail cleanly either way
        pass
    c = C.c_void_p()
    rc = cs.lib().crn_comm_create(0, 0, 1, (C.c_uint8 * 128)(), 64, 2, C.byref(c))
    assert rc < 0 and not c.value
    assert cs.lib().crn_comm_create(0, 3, 2, (C.c_uint8 * 128)(), 64, 2, C.byref(c)) == -1   # rank >= world


@pytest.mark.gpu
def test_one_rank_communicator_end_to_end(built):
    import torch
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    stream = torch.cuda.current_stream().cuda_stream
    cfg = cs.cfg_reference()
    E = 64
    comm = cs.Comm(0, 0, 1, cs.comm_unique_id(), E * cfg.n_bands, depth=2)
    sensor = cs.Sensor(cfg)
    spe = cs.samples_per_epoch(cfg)
    iq = torch.zeros(E * spe * 2, dtype=torch.float32, device=dev)
    truth = torch.empty(E, dtype=torch.int32, device=dev)
    dec = torch.empty(E, dtype=torch.int32, device=dev)
    ptrs = set()
    for step in range(5):                      # more steps than slots: both slots are reused
        sensor.synth_fill_device(iq.data_ptr(), E, spe, seed=100 + step, truth_ptr=truth.data_ptr(), stream=stream)
        occ_ptr = comm.local(step, stream)     # makes `stream` wait for this slot's previous gather
        ptrs.add(occ_ptr)
        sensor.run_device(iq.data_ptr(), E, 512, {"features": 0, "ann_out": 0, "decision": dec.data_ptr(),
                                                  "occupancy": occ_ptr, "spectrum": 0}, stream=stream)
        comm.allgather(step, stream)           # only enqueues: side stream, behind an event of `stream`
        comm.finish(stream)
        torch.cuda.synchronize()
        got = np.frombuffer(cs.device_to_host(comm.gathered(step), E * cfg.n_bands), np.uint8).reshape(E, cfg.n_bands)
        picks = truth.cpu().numpy()
        want = np.zeros((E, cfg.n_bands), np.uint8)
        want[np.nonzero(picks)[0], picks[picks > 0]] = 1     # one-hot over bands 1..3 from the decision
        assert np.array_equal(got, want), step
        assert np.array_equal(dec.cpu().numpy(), picks)
    assert len(ptrs) == 2
    info = comm.info()                         # asked of RCCL (ncclCommCount / UserRank / CuDevice / GetVersion), not echoed
    assert info["nranks"] == 1 and info["rank"] == 0 and info["rccl_device"] == 0 and info["rccl_version"] >= 20000
    assert info["gathers"] == 5 and info["depth"] == 2 and info["bytes_per_rank"] == E * cfg.n_bands and "rccl" in info["library"]
    with pytest.raises(cs.CrnError):
        comm.local(-1, stream)
    sensor.close()
    comm.close()


def _run_group(cmd, timeout, env, cwd=None, attempts=2):
    """Run cmd in its own process group; on a timeout (a communicator bring-up that hangs on a bad box has been seen once) kill the
    whole group — children a launcher script put in the background included — and try once more."""
    import signal
    import subprocess
    for attempt in range(attempts):
        p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, cwd=cwd, start_new_session=True)
        try:
            out, err = p.communicate(timeout=timeout)
            return subprocess.CompletedProcess(cmd, p.returncode, out, err)
        except subprocess.TimeoutExpired:
            os.killpg(p.pid, signal.SIGKILL)
            p.communicate()
            if attempt + 1 == attempts:
                raise


@pytest.mark.gpu
def test_scan_node_cpp_program_over_the_c_abi(built, tmp_path):
    """BASELINE.json configs[4] as a C++ host program that uses nothing but include/crn_sense.h (tests/harness/scan_node.cpp):
    streams of 64 Welch channels generated on the device (Markov primary user), sensed, occupancy gathered over RCCL through
    crn_comm_* — here as a world of one rank (the pool hands out one GPU).  Exit code 0 = own block in place in the gathered
    vector and exactly the driven channel occupied in every epoch."""
    import os
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "harness", "scan_node")
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    out = _run_group([exe, "4", "256", "10", str(tmp_path / "rccl_id")], 120, env)
    assert out.returncode == 0, out.stdout + out.stderr
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("scan_node rank 0/1")][0]
    assert "own block in place: yes" in line and "driven channel flagged in 1024 of 1024 epochs" in line
    comm_line = [ln for ln in out.stdout.splitlines() if ln.startswith("scan_node: RCCL communicator:")][0]
    assert "1 ranks, rank 0 on device 0" in comm_line and "version 2" in comm_line      # real RCCL 2.x, asked of the communicator itself
    occupied = float(line.split(" epochs, ")[-1].split()[0])
    assert 1.0 <= occupied < 8.0       # the driven channel, plus splatter where the traffic changes inside a frame
    print(line)


@pytest.mark.gpu
def test_scan_node_two_processes_on_one_gpu(built):
    """cfg4 without Python as TWO processes (tools/run_scan_node.sh 2, both on the box's one GPU) over the shared-memory stand-in for
    RCCL: unique id through a file, streams split between the ranks, thresholds calibrated per rank, every rank's block found at its
    place in the gathered vector (the program checks that itself and exits non-zero otherwise)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, ONE_GPU="1", CRN_RCCL_LIB=os.path.join(root, "tests", "harness", "libfake_rccl_mp.so"))
    out = _run_group(["bash", os.path.join(root, "tools", "run_scan_node.sh"), "2", "8", "64", "5"], 120, env, cwd=root)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-1500:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("scan_node rank")]
    assert len(lines) == 2 and any("rank 0/2" in ln for ln in lines) and any("rank 1/2" in ln for ln in lines), out.stdout
    assert "scan_node: RCCL communicator: 2 ranks" in out.stdout


@pytest.mark.gpu
def test_scan_node_ranks_stop_together_when_one_fails_its_setup(built, tmp_path):
    """crn_comm_create is collective and cannot report a peer's local failure: tests/harness/scan_node.cpp therefore has its ranks agree
    (files beside the id file) that every one finished its local set-up before any of them enters the collective.  Two ranks, one of
    which fails its set-up (a device ordinal that does not exist): BOTH exit non-zero within seconds — the healthy one is not left
    waiting inside ncclCommInitRank."""
    import subprocess
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "tests", "harness", "scan_node")
    fake = os.path.join(root, "tests", "harness", "libfake_rccl_mp.so")
    idf = str(tmp_path / "id")
    t0 = time.time()
    procs = []
    for r, dev in ((0, 0), (1, 99)):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(dev), CRN_RCCL_LIB=fake, HIP_VISIBLE_DEVICES="0")
        procs.append(subprocess.Popen([exe, "8", "32", "2", idf], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=150) for p in procs]
    assert all(p.returncode != 0 for p in procs), [p.returncode for p in procs]
    assert time.time() - t0 < 120
    assert "failed its set-up" in outs[0][1] and "stopping before the collective" in outs[0][1]

