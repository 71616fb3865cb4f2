"""north_star's ">= 70 % of the HBM-read roofline" as a tested property: bench.py on the driver's shape (`--steps 20 --warmup 5`, default
batch) must reach the floors of tests/perf_floors.py on every leg — headline, SURVEY §8(d) cfgH's 2 GiB batch on one and on two streams,
the unpruned kernel, and the cfg1 / cfg2 / cfg3 configurations — and the kernel time must fit inside the step time.

(The file sorts last on purpose: the driver runs the GPU suite with -x, and a throughput gate should be the last thing that can stop it.)

$CRN_FLOOR_EXTRA_ARGS (tools/gpu_floor_gate.sh only) appends arguments to every bench.py call, to show the gate going red on a deliberately
bad configuration; $CRN_SENSE_LIB selects another build of the library as everywhere else.
"""
import json
import os
import shlex
import subprocess
import sys

import pytest

import perf_floors as pf

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu
EXTRA = shlex.split(os.environ.get("CRN_FLOOR_EXTRA_ARGS", ""))
_log = []


def _bench(*args):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), *pf.DRIVER_SHAPE, "--cpu-epochs", "0", "--no-live-traffic", *args, *EXTRA]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def _fracs(d):
    out = {"main": d["roofline"]["frac"]}
    for k, leg in (d["config"].get("alt") or {}).items():
        out["alt." + k] = leg["frac"]
    return out


def _best_of(args, want):
    """Run bench.py up to ATTEMPTS times (fresh process each) until every leg in `want` ({leg: floor}) is at or above its floor; returns
    (the passing or last line, [fractions of every attempt])."""
    seen, d = [], None
    for _ in range(pf.ATTEMPTS):
        d = _bench(*args)
        f = _fracs(d)
        seen.append(f)
        if all(f.get(leg, 0.0) >= floor for leg, floor in want.items()):
            break
    return d, seen


def _record(name, want, seen):
    _log.append({"leg": name, "floors": want, "attempts": seen})
    d = os.environ.get("CRN_EVIDENCE_DIR")
    if d and os.path.isdir(d):
        with open(os.path.join(d, "roofline_floors.json"), "w") as f:
            json.dump({"shape": "bench.py " + " ".join(pf.DRIVER_SHAPE) + " --cpu-epochs 0 --no-live-traffic " + " ".join(EXTRA),
                       "library": os.environ.get("CRN_SENSE_LIB", "libcrnsense.so"), "legs": _log}, f, indent=1)


def _check(name, d, seen, want):
    last = seen[-1]
    for leg, floor in want.items():
        assert leg in last, f"{name}: bench.py's line has no leg '{leg}' (legs: {sorted(last)})"
        assert last[leg] >= floor, (f"{name}: {leg} reached {last[leg]:.4f} of the HBM roofline, floor {floor} (tests/perf_floors.py); "
                                    f"{len(seen)} attempts: {[round(s.get(leg, 0.0), 4) for s in seen]}")
        assert last[leg] < 1.0
    r = d["roofline"]
    assert r["kernel_ms_mean"] <= d["ms_per_step"], (r["kernel_ms_mean"], d["ms_per_step"])     # kernel time fits inside the driver-visible step
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and r["peak"] == 8000.0


def test_headline_and_its_alt_legs_hold_their_floors(built):
    """The driver's own command.  The headline also has to clear north_star's 70 % with the floor's margin on top."""
    want = {"main": pf.FLOORS["headline"], **{k: v for k, v in pf.FLOORS.items() if k.startswith("alt.")}}
    assert pf.FLOORS["headline"] >= pf.NORTH_STAR and pf.FLOORS["alt.cfgH_2GiB_batch"] >= pf.NORTH_STAR
    d, seen = _best_of((), want)
    _record("headline", want, seen)
    _check("headline", d, seen, want)
    assert d["config"]["epochs_per_gpu"] == 28672 and d["config"]["fft_len"] == 4096 and d["steps"] == 20 and d["warmup"] == 5
    assert d["config"]["cfgH_as_worded_frac"] == d["config"]["alt"]["cfgH_2GiB_batch"]["frac"]
    # Msamples/s follows from the wall clock of the same 20 steps: never above what the kernel time allows
    assert d["value"] * 8e6 / 1e9 <= d["roofline"]["achieved"] * 1.0001


@pytest.mark.parametrize("name,args", [("ref512", ("--mode", "ref")), ("energy1024", ("--fft", "1024")), ("welch4096", ("--mode", "welch"))])
def test_other_configurations_hold_their_floors(built, name, args):
    want = {"main": pf.FLOORS[name]}
    d, seen = _best_of((*args, "--no-alt"), want)
    _record(name, want, seen)
    _check(name, d, seen, want)
