"""The C-ABI library loads on a GPU-less host, exports everything include/crn_sense.h declares,
and fails loudly (no CPU fallback) when asked to compute without a device."""
import ctypes as C
import os
import re
import subprocess

import pytest

import crnsense as cs

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(text):
    return set(re.findall(r"\b(crn_[a-z0-9_]+)\s*\(", text))


def _exported(path):
    out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
    return {ln.split()[-1] for ln in out.splitlines() if " T " in ln}


def test_header_symbols_are_exported(built):
    """Every include/*.h against the library that implements it: libcrnsense.so exports exactly what include/crn_sense.h declares (no
    more: nothing undeclared leaks out of the product; no less); the optional libcrnsense_sc16.so (make SC16=1) and the test artefact
    libcrnsense_plain.so export that plus exactly the five entry points of include/crn_sense_sc16.h; libcrnliquidfft.so exports
    include/crn_liquid_fft.h's three."""
    inc = os.path.join(ROOT, "include")
    assert sorted(os.listdir(inc)) == ["crn_liquid_fft.h", "crn_sense.h", "crn_sense_sc16.h"]
    hdr = open(os.path.join(inc, "crn_sense.h")).read()
    assert "#if" not in hdr.replace("#ifndef CRN_SENSE_H", "").replace("#ifdef __cplusplus", "").replace("#if defined(__GNUC__)", "")   # no optional blocks
    declared, optional = _declared(hdr), _declared(open(os.path.join(inc, "crn_sense_sc16.h")).read())
    assert declared == set(cs.EXPORTS), declared ^ set(cs.EXPORTS)
    assert optional == set(cs.SC16_EXPORTS), optional ^ set(cs.SC16_EXPORTS)
    L = cs.lib()
    for name in declared:
        assert hasattr(L, name), name
    assert L.crn_abi_version() == cs.CRN_ABI_VERSION
    if not os.environ.get("CRN_SENSE_LIB"):
        assert _exported(cs.LIB_PATH) == declared, _exported(cs.LIB_PATH) ^ declared
    for path in (cs.SC16_LIB_PATH, cs.PLAIN_LIB_PATH):
        if os.path.exists(path):
            assert _exported(path) == declared | optional, path
    liquid = set(re.findall(r"\b(fft_[a-z_]+)\s*\(", open(os.path.join(inc, "crn_liquid_fft.h")).read()))
    assert liquid == {"fft_create_plan", "fft_execute", "fft_destroy_plan"} and liquid <= _exported(cs.LIQUID_SHIM_PATH)


def test_engine_directory_is_self_contained(built):
    """a CRTS tree copies cognitive_engines/CE_Predictive_Node_GPU/ as it is, so the directory
    carries CE_X.cpp + CE_X.hpp (what src/config_cognitive_engines.cpp:43-61 requires) and its own, identical,
    copy of the C ABI header — and no test double."""
    d = os.path.join(ROOT, "cognitive-radio-network_amd", "cognitive_engines", "CE_Predictive_Node_GPU")
    assert sorted(os.listdir(d)) == ["CE_Predictive_Node_GPU.cpp", "CE_Predictive_Node_GPU.hpp", "crn_sense.h"]
    assert open(os.path.join(d, "crn_sense.h")).read() == open(os.path.join(ROOT, "include", "crn_sense.h")).read()


def test_cfg_struct_layout_round_trips(built):
    c = cs.cfg_energy_scaled(4096, 4.0)
    assert (c.abi_version, c.fft_len, c.frames_per_epoch, c.hop) == (cs.CRN_ABI_VERSION, 4096, 10, 4096)
    assert (c.mode, c.decide, c.window, c.n_bands, c.n_segs, c.ref_band) == (1, 1, 0, 4, 5, 0)
    segs = [(c.segs[i].lo, c.segs[i].hi, c.segs[i].band) for i in range(5)]
    assert segs == [(0, 128, 1), (3968, 4088, 1), (440, 680, 2), (1512, 1776, 3), (2400, 2480, 0)]
    assert c.thresh[0] == float("inf")
    assert abs(c.thresh[1] - 4.0 * 248 / 80) < 1e-6
    assert c.ann_threshold == 0.8  # tail of the struct: catches any padding mismatch
    w = cs.cfg_welch(4096, 8, 64)
    assert (w.hop, w.window, w.n_bands, w.n_segs, w.ref_band) == (2048, 1, 64, 64, -1)
    assert (w.segs[63].lo, w.segs[63].hi, w.segs[63].band) == (4032, 4096, 63)


def test_bad_configs_are_rejected(built):
    L = cs.lib()
    c = cs.Cfg()
    assert L.crn_cfg_energy_scaled(C.byref(c), 1000, 4.0) == -1
    assert b"multiple of 512" in L.crn_last_error()
    assert L.crn_cfg_welch(C.byref(c), 4096, 8, 63) == -1
    h = C.c_void_p()
    bad = cs.cfg_reference()
    bad.fft_len = 500
    assert L.crn_sense_create(C.byref(bad), C.byref(h)) == -1
    bad = cs.cfg_reference()
    bad.segs[1].hi = 513
    assert L.crn_sense_create(C.byref(bad), C.byref(h)) == -1
    assert b"segment 1" in L.crn_last_error()


def test_no_cpu_fallback_without_gpu(built):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(cs.CrnError) as ei:
        cs.Sensor(cs.cfg_reference())
    assert "-2" in str(ei.value) or "device" in str(ei.value).lower()


def _documented_valid(c):
    """include/crn_sense.h, crn_cfg: the constraints crn_sense_create documents, restated."""
    if c.abi_version != cs.CRN_ABI_VERSION or c.fft_len not in (512, 1024, 2048, 4096):
        return False
    if c.frames_per_epoch < 1 or not (1 <= c.hop <= c.fft_len):
        return False
    if c.mode not in (0, 1) or c.decide not in (0, 1, 2) or c.window not in (0, 1, 2):
        return False
    if not (1 <= c.n_bands <= cs.CRN_MAX_BANDS) or not (1 <= c.n_segs <= cs.CRN_MAX_SEGS):
        return False
    for i in range(c.n_segs):
        g = c.segs[i]
        if g.lo < 0 or g.hi > c.fft_len or g.lo > g.hi or not (0 <= g.band < c.n_bands):
            return False
    if c.decide == cs.DECIDE_ANN and c.n_bands != 4:
        return False
    if c.decide == cs.DECIDE_THRESHOLD and c.ref_band >= c.n_bands:
        return False
    return True


def test_config_validation_fuzz(built):
    """Random edits of valid configurations (edge values in every integer field, segments moved out of range): crn_sense_create
    refuses exactly the ones the header documents as invalid with CRN_ERR_ARG before it touches the device, never crashes, and
    never returns a handle for them.  (Without a GPU a valid configuration fails later, with CRN_ERR_DEVICE.)"""
    import torch
    from hypothesis import given, settings, strategies as st
    has_gpu = torch.cuda.is_available()
    L = cs.lib()
    edge = st.sampled_from([-2 ** 31, -1, 0, 1, 2, 3, 4, 5, 64, 79, 80, 81, 160, 161, 511, 512, 513, 1024, 2048, 4095, 4096, 4097, 8192, 2 ** 31 - 1])
    fields = ["abi_version", "fft_len", "frames_per_epoch", "hop", "mode", "decide", "window", "n_bands", "n_segs", "ref_band"]
    bases = [cs.cfg_reference, lambda: cs.cfg_energy_scaled(4096, 4.0), lambda: cs.cfg_welch(4096, 8, 64), lambda: cs.cfg_welch(1024, 8, 64)]

    @settings(max_examples=400, deadline=None)
    @given(base=st.integers(0, len(bases) - 1),
           edits=st.lists(st.tuples(st.sampled_from(fields), edge), min_size=0, max_size=3),
           seg_edits=st.lists(st.tuples(st.integers(0, cs.CRN_MAX_SEGS - 1), st.sampled_from(["lo", "hi", "band"]), edge), min_size=0, max_size=2))
    def check(base, edits, seg_edits):
        c = bases[base]()
        for f, v in edits:
            setattr(c, f, v)
        for i, f, v in seg_edits:
            setattr(c.segs[i], f, v)
        h = C.c_void_p()
        rc = L.crn_sense_create(C.byref(c), C.byref(h))
        if not _documented_valid(c):
            assert rc == cs.CRN_ERR_ARG and not h.value, (rc, L.crn_last_error())
        elif has_gpu:
            # large-but-valid shapes may be refused for memory, never accepted wrongly
            assert rc in (0, -3), (rc, L.crn_last_error())
            if rc == 0:
                L.crn_sense_destroy(h)
        else:
            assert rc == -2 and not h.value, (rc, L.crn_last_error())

    check()
