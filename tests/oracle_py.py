"""ctypes access to oracle/libcrn_oracle.so — the CPU restatement of the reference path.

Test infrastructure: imported only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline.
"""
import ctypes as C
import os
import subprocess

import numpy as np

import crnsense as cs

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_LIB = os.environ.get("CRN_ORACLE_LIB") or os.path.join(ORACLE_DIR, "libcrn_oracle.so")  # env: sanitizer build

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(ORACLE_LIB):
            subprocess.check_call(["make", "-C", ORACLE_DIR, ORACLE_LIB])
        L = C.CDLL(ORACLE_LIB)
        L.crn_oracle_fft_radix2.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.crn_oracle_dft_f64.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.crn_oracle_ann.argtypes = [C.c_void_p, C.c_void_p]
        L.crn_oracle_ref_epoch.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.crn_oracle_ref_weights.argtypes = [C.c_void_p, C.c_void_p]
        L.crn_oracle_run.argtypes = [C.POINTER(cs.Cfg), C.c_void_p, C.c_int64, C.c_int32, C.c_int64,
                                     C.POINTER(cs.Out), C.c_int32]
        L.crn_oracle_synth.argtypes = [C.POINTER(cs.Cfg), C.POINTER(cs.SynthCfg), C.c_void_p, C.c_int64, C.c_int64,
                                       C.c_void_p]
        L.crn_oracle_ann_train.argtypes = [C.POINTER(cs.TrainCfg), C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p,
                                           C.c_void_p, C.POINTER(C.c_double)]
        _lib = L
    return _lib


def use_native_build():
    """bench.py's cpu_baseline leg: switch to a -O2 -march=native build of the same sources, compiled here and now
    (oracle/_native/, `make -C oracle native`).  Returns False — and keeps the portable build — if that fails."""
    global _lib, ORACLE_LIB
    path = os.path.join(ORACLE_DIR, "_native", "libcrn_oracle_native.so")
    try:
        subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "native"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    except Exception:
        return False
    if not os.path.exists(path):
        return False
    ORACLE_LIB = path
    _lib = None
    try:
        lib()
    except Exception:
        ORACLE_LIB = os.path.join(ORACLE_DIR, "libcrn_oracle.so")
        _lib = None
        return False
    return True


def fft_radix2(x):
    """x: complex64 [N] -> complex64 [N] (liquid-dsp style fp32 radix-2 DIT)."""
    x = np.ascontiguousarray(x, dtype=np.complex64)
    y = np.empty_like(x)
    assert lib().crn_oracle_fft_radix2(x.ctypes.data, y.ctypes.data, x.size) == 0
    return y


def dft_f64(x):
    x = np.ascontiguousarray(x, dtype=np.complex128)
    y = np.empty_like(x)
    assert lib().crn_oracle_dft_f64(x.ctypes.data, y.ctypes.data, x.size) == 0
    return y


def ann(feat4):
    f = np.ascontiguousarray(feat4, dtype=np.float32)
    o = np.zeros(3, np.float64)
    d = lib().crn_oracle_ann(f.ctypes.data, o.ctypes.data)
    return d, o


def ref_weights():
    wih = np.zeros((5, 6), np.float64)
    who = np.zeros((6, 4), np.float64)
    lib().crn_oracle_ref_weights(wih.ctypes.data, who.ctypes.data)
    return wih, who


def ref_epoch(iq, L):
    """iq: float32 interleaved [10*L*2] (10 packets of L samples). Literal reference epoch."""
    iq = np.ascontiguousarray(iq, dtype=np.float32)
    assert iq.size == 10 * L * 2
    avg = np.zeros(512, np.float32)
    feat = np.zeros(4, np.float32)
    out3 = np.zeros(3, np.float64)
    tx = C.c_double(0.0)
    d = lib().crn_oracle_ref_epoch(iq.ctypes.data, L, avg.ctypes.data, feat.ctypes.data, out3.ctypes.data,
                                   C.byref(tx))
    assert d >= 0
    return {"decision": d, "fft_avg": avg, "features": feat, "ann_out": out3, "tx_freq": tx.value}


def run(cfg, iq, n_epochs, L=None, want_spectrum=False, n_threads=1, epoch_stride=0):
    """Generalised oracle pipeline; same outputs as crnsense.Sensor.run_host."""
    L = cfg.fft_len if L is None else L
    iq = np.ascontiguousarray(iq, dtype=np.float32)
    res = {
        "features": np.zeros((n_epochs, cfg.n_bands), np.float32),
        "ann_out": np.zeros((n_epochs, 3), np.float64),
        "decision": np.zeros((n_epochs,), np.int32),
        "occupancy": np.zeros((n_epochs, cfg.n_bands), np.uint8),
    }
    if want_spectrum:
        res["spectrum"] = np.zeros((n_epochs, cfg.fft_len), np.float32)
    o = cs.Out(**{k: v.ctypes.data for k, v in res.items()})
    rc = lib().crn_oracle_run(C.byref(cfg), iq.ctypes.data, n_epochs, L, epoch_stride, C.byref(o), n_threads)
    assert rc == 0, "crn_oracle_run rejected the configuration"
    return res


def synth(cfg, sc, n_epochs, spe):
    """CPU twin of crn_synth_fill_device_ex: returns (iq float32 [n_epochs * spe * 2], truth int32 [n_epochs])."""
    iq = np.zeros(n_epochs * spe * 2, np.float32)
    truth = np.zeros(n_epochs, np.int32)
    rc = lib().crn_oracle_synth(C.byref(cfg), C.byref(sc), iq.ctypes.data, n_epochs, spe, truth.ctypes.data)
    assert rc == 0, rc
    return iq, truth


def ann_train(tc, features, labels):
    """CPU twin of crn_ann_train_device: returns (w_ih [5][6], w_ho [6][4], loss)."""
    f = np.ascontiguousarray(features, dtype=np.float32)
    lab = np.ascontiguousarray(labels, dtype=np.int32)
    wih, who = np.zeros((5, 6), np.float64), np.zeros((6, 4), np.float64)
    loss = C.c_double()
    rc = lib().crn_oracle_ann_train(C.byref(tc), f.ctypes.data, lab.ctypes.data, f.shape[0], wih.ctypes.data,
                                    who.ctypes.data, C.byref(loss))
    assert rc == 0, rc
    return wih, who, loss.value
