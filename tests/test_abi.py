"""The C-ABI library loads on a GPU-less host, exports everything include/crn_sense.h declares,
and fails loudly (no CPU fallback) when asked to compute without a device."""
import ctypes as C
import os
import re

import pytest

import crnsense as cs

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_are_exported(built):
    hdr = open(os.path.join(ROOT, "include", "crn_sense.h")).read()
    declared = set(re.findall(r"\b(crn_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(cs.EXPORTS), declared ^ set(cs.EXPORTS)
    L = cs.lib()
    for name in declared:
        assert hasattr(L, name), name
    assert L.crn_abi_version() == cs.CRN_ABI_VERSION


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
