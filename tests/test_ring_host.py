"""The ingest ring's HOST logic without a GPU: csrc/crn_ingest.cpp compiled against a host-only stand-in for the HIP
runtime (tests/harness/fake_hip) and run under ThreadSanitizer (tests/harness/ring_unit.cpp): slot hand-out and single
copy, the hand-off between the caller's thread and the ring's launcher thread, BUSY refusals with a slow "GPU" (push never
waits), uneven stream rates (open epochs carried to the other buffer), flush in the middle of epochs, packet-length
changes, and a launch that fails on the launcher thread — every (stream, epoch) exactly once, in order, carrying the
checksum of exactly its own ten packets, and no data race reported."""
import os
import subprocess

EXE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "harness", "ring_unit")


def test_ring_host_logic_under_thread_sanitizer(built):
    if not os.path.exists(EXE):
        subprocess.check_call(["make", "-C", os.path.dirname(EXE), EXE])
    for _ in range(5):   # thread interleavings differ from run to run
        out = subprocess.run([EXE], capture_output=True, text=True, timeout=120)
        assert out.returncode == 0, out.stdout + out.stderr
        assert "ring_unit: ok" in out.stdout
        assert "ThreadSanitizer" not in out.stderr, out.stderr[:3000]
