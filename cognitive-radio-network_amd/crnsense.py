"""ctypes binding of libcrnsense's C ABI (include/crn_sense.h).

Plumbing for tests/, bench.py and __graft_entry__.py only: the product is the C-ABI shared
library and the C++ engine in cognitive_engines/CE_Predictive_Node_GPU/; nothing here computes.  There is no CPU fallback: loading
fails loudly when libcrnsense.so has not been built, and crn_sense_create fails when no GPU is
visible.
"""
import ctypes as C
import os
import weakref

HERE = os.path.dirname(os.path.abspath(__file__))
# env overrides: CRN_SENSE_LIB = another build of the library (libcrnsense_sc16.so: the optional wire-format kernels, make SC16=1;
# libcrnsense_plain.so: no assembly filter, a test artefact); CRN_SENSE_AB=1 = libcrnsense_ab.so, the build that also carries the
# measurement variants (tools/, the A/B test) — the engine and bench.py's default run use the shipped library
LIB_PATH = os.environ.get("CRN_SENSE_LIB") or os.path.join(HERE, "libcrnsense_ab.so" if os.environ.get("CRN_SENSE_AB") == "1" else "libcrnsense.so")
SC16_LIB_PATH = os.path.join(HERE, "libcrnsense_sc16.so")
PLAIN_LIB_PATH = os.path.join(HERE, "libcrnsense_plain.so")
LIQUID_SHIM_PATH = os.path.join(HERE, "libcrnliquidfft.so")  # include/crn_liquid_fft.h

CRN_ABI_VERSION = 4
CRN_ERR_ARG = -1
CRN_ERR_BUSY = -5
CRN_MAX_BANDS = 80
CRN_MAX_SEGS = 160

MODE_REF_MAG, MODE_ENERGY = 0, 1
DECIDE_ANN, DECIDE_THRESHOLD, DECIDE_NONE = 0, 1, 2
WINDOW_RECT, WINDOW_HANN, WINDOW_BLACKMAN_HARRIS = 0, 1, 2

# the optional wire-format entry points (include/crn_sense_sc16.h): only in a library built with make SC16=1
SC16_EXPORTS = ["crn_sense_run_device_sc16", "crn_pack_sc16_device", "crn_sense_set_wire_full_scale", "crn_ingest_create_sc16", "crn_ingest_push_sc16"]

# every symbol include/crn_sense.h declares unconditionally (tests check the library exports them all)
EXPORTS = [
    "crn_cfg_reference", "crn_cfg_energy_scaled", "crn_cfg_welch", "crn_cfg_reference_scaled", "crn_cfg_welch_scaled",
    "crn_cfg_save_ann", "crn_cfg_load_ann", "crn_sense_set_ann", "crn_sense_set_bands", "crn_noise_floor_host", "crn_sense_synchronize",
    "crn_sense_create", "crn_sense_destroy", "crn_sense_run_device", "crn_sense_run_host",
    "crn_synth_fill_device", "crn_synth_fill_device_ex", "crn_ann_train_device", "crn_fft_forward_device",
    "crn_sense_kernel_info", "crn_sense_set_variant", "crn_sense_dealt_launches",
    "crn_ingest_create", "crn_ingest_push", "crn_ingest_flush", "crn_ingest_poll", "crn_ingest_drain",
    "crn_ingest_destroy", "crn_ingest_set_packet_len", "crn_ingest_wait", "crn_ingest_dropped", "crn_ingest_packets_per_epoch",
    "crn_noise_floor_device", "crn_sense_set_thresholds", "crn_sense_reserve_noise_floor", "crn_sense_calibrate_thresholds",
    "crn_ingest_calibrate", "crn_ingest_noise_floor",
    "crn_sense_reserve_host", "crn_sense_set_timing", "crn_sense_get_stats", "crn_ingest_get_stats",
    "crn_monitor_rows_device",
    "crn_comm_unique_id", "crn_comm_create", "crn_comm_local", "crn_comm_allgather", "crn_comm_gathered",
    "crn_comm_finish", "crn_comm_destroy", "crn_comm_local_addr", "crn_comm_wait", "crn_comm_info",
    "crn_last_error", "crn_abi_version", "crn_build_info",
]


class BandSeg(C.Structure):
    _fields_ = [("lo", C.c_int32), ("hi", C.c_int32), ("band", C.c_int32)]


class Cfg(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32), ("fft_len", C.c_int32), ("frames_per_epoch", C.c_int32),
        ("hop", C.c_int32), ("mode", C.c_int32), ("decide", C.c_int32), ("window", C.c_int32),
        ("n_bands", C.c_int32), ("n_segs", C.c_int32), ("ref_band", C.c_int32),
        ("device", C.c_int32), ("reserved0", C.c_int32),
        ("segs", BandSeg * CRN_MAX_SEGS),
        ("thresh", C.c_float * CRN_MAX_BANDS),
        ("ann_w_ih", (C.c_double * 6) * 5),
        ("ann_w_ho", (C.c_double * 4) * 6),
        ("ann_threshold", C.c_double),
        ("tx_freq_for_decision", C.c_double * 4),
    ]


class Out(C.Structure):
    _fields_ = [("features", C.c_void_p), ("ann_out", C.c_void_p), ("decision", C.c_void_p),
                ("occupancy", C.c_void_p), ("spectrum", C.c_void_p)]


class EpochResult(C.Structure):
    _fields_ = [("stream", C.c_int32), ("decision", C.c_int32), ("epoch_seq", C.c_int64),
                ("ann_out", C.c_double * 3), ("features", C.c_float * CRN_MAX_BANDS),
                ("occupancy", C.c_uint8 * CRN_MAX_BANDS), ("flags", C.c_int32), ("noise_floor", C.c_float)]


EPOCH_CALIBRATION = 1


MONITOR_GNURADIO, MONITOR_PSD = 0, 1
PU_UNIFORM, PU_MARKOV_AS_WRITTEN, PU_MARKOV_INTENDED, PU_SWEEP = 0, 1, 2, 3
SIG_TONES, SIG_CW, SIG_BAND_NOISE, SIG_RRC_QPSK, SIG_GMSK, SIG_OFDM = 0, 1, 2, 3, 4, 5


class SenseStats(C.Structure):
    _fields_ = [("launches", C.c_int64), ("epochs", C.c_int64), ("samples", C.c_int64), ("timed_launches", C.c_int64),
                ("kernel_ms", C.c_double), ("kernel_ms_last", C.c_double), ("kernel_ms_min", C.c_double), ("kernel_ms_max", C.c_double)]


class IngestStats(C.Structure):
    _fields_ = [("packets", C.c_int64), ("dropped", C.c_int64), ("batches", C.c_int64), ("batches_failed", C.c_int64),
                ("epochs_launched", C.c_int64), ("epochs_ready", C.c_int64), ("epochs_polled", C.c_int64),
                ("latency_us_sum", C.c_double), ("latency_us_max", C.c_double)]


class CommInfo(C.Structure):
    _fields_ = [("nranks", C.c_int32), ("rank", C.c_int32), ("rccl_device", C.c_int32), ("rccl_version", C.c_int32),
                ("device", C.c_int32), ("depth", C.c_int32), ("bytes_per_rank", C.c_int64), ("gathers", C.c_int64),
                ("library", C.c_char * 128), ("pci_bus_id", C.c_char * 32)]


class SynthCfg(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("noise_power", C.c_float), ("signal_rms", C.c_float),
                ("tones_per_band", C.c_int32), ("pu_model", C.c_int32), ("signal_kind", C.c_int32),
                ("n_streams", C.c_int32), ("adc_bits", C.c_int32)]


class TrainCfg(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("iterations", C.c_int32), ("restarts", C.c_int32), ("eta", C.c_float),
                ("alpha", C.c_float), ("normalise", C.c_int32), ("reserved", C.c_int32)]


class CrnError(RuntimeError):
    pass


_lib = None


def lib():
    """Load libcrnsense.so (built by __graft_entry__.build() / csrc/Makefile)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise CrnError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(there is no CPU fallback)")
        # A process that uses both PyTorch and this library must load PyTorch first: the wheel bundles its own ROCm runtime, and when
        # /opt/rocm's copy (which libcrnsense links) is loaded before it the two HSA runtimes collide — the library then reports
        # "no ROCm-capable device is detected".  Tests and bench.py use torch for device memory, so pin the order here.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        if L.crn_abi_version() != CRN_ABI_VERSION:   # the ctypes structures below mirror ONE version of include/crn_sense.h
            raise CrnError(f"{LIB_PATH} speaks ABI version {L.crn_abi_version()}, crnsense.py mirrors version {CRN_ABI_VERSION}: rebuild the library")
        L.crn_last_error.restype = C.c_char_p
        L.crn_build_info.argtypes = [C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        L.crn_cfg_reference.argtypes = [C.POINTER(Cfg)]
        L.crn_cfg_energy_scaled.argtypes = [C.POINTER(Cfg), C.c_int32, C.c_float]
        L.crn_cfg_welch.argtypes = [C.POINTER(Cfg), C.c_int32, C.c_int32, C.c_int32]
        L.crn_cfg_reference_scaled.argtypes = [C.POINTER(Cfg), C.c_int32]
        L.crn_cfg_welch_scaled.argtypes = [C.POINTER(Cfg), C.c_int32, C.c_int32, C.c_float]
        L.crn_cfg_save_ann.argtypes = [C.POINTER(Cfg), C.c_char_p]
        L.crn_cfg_load_ann.argtypes = [C.POINTER(Cfg), C.c_char_p]
        L.crn_sense_set_ann.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p]
        L.crn_sense_set_bands.argtypes = [C.c_void_p, C.POINTER(BandSeg), C.c_int32, C.c_int32, C.POINTER(C.c_float)]
        L.crn_sense_synchronize.argtypes = [C.c_void_p, C.c_void_p]
        L.crn_noise_floor_host.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_float)]
        L.crn_sense_create.argtypes = [C.POINTER(Cfg), C.POINTER(C.c_void_p)]
        L.crn_sense_destroy.argtypes = [C.c_void_p]
        L.crn_sense_run_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int64,
                                           C.POINTER(Out), C.c_void_p]
        if hasattr(L, "crn_sense_run_device_sc16"):   # a library built with make SC16=1
            L.crn_sense_run_device_sc16.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int64,
                                                    C.POINTER(Out), C.c_void_p]
            L.crn_sense_set_wire_full_scale.argtypes = [C.c_void_p, C.c_double]
            L.crn_pack_sc16_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
            L.crn_ingest_create_sc16.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]
            L.crn_ingest_push_sc16.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
        L.crn_sense_run_host.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int64,
                                         C.POINTER(Out)]
        L.crn_synth_fill_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_uint64,
                                            C.c_float, C.c_float, C.c_int32, C.c_void_p, C.c_void_p]
        L.crn_synth_fill_device_ex.argtypes = [C.c_void_p, C.POINTER(SynthCfg), C.c_void_p, C.c_int64, C.c_int64,
                                               C.c_void_p, C.c_void_p]
        L.crn_ann_train_device.argtypes = [C.c_void_p, C.POINTER(TrainCfg), C.c_void_p, C.c_void_p, C.c_int64,
                                           C.c_void_p, C.c_void_p, C.POINTER(C.c_double), C.c_void_p]
        L.crn_fft_forward_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int64, C.c_void_p,
                                             C.c_void_p]
        L.crn_sense_kernel_info.argtypes = [C.c_void_p, C.c_char_p, C.c_int32, C.POINTER(C.c_int32),
                                            C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        L.crn_sense_set_variant.argtypes = [C.c_void_p, C.c_int32]
        L.crn_ingest_create.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]
        L.crn_ingest_push.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
        L.crn_ingest_flush.argtypes = [C.c_void_p]
        L.crn_ingest_poll.argtypes = [C.c_void_p, C.POINTER(EpochResult), C.c_int32, C.POINTER(C.c_int32)]
        L.crn_ingest_drain.argtypes = [C.c_void_p]
        L.crn_ingest_destroy.argtypes = [C.c_void_p]
        L.crn_ingest_set_packet_len.argtypes = [C.c_void_p, C.c_int32]
        L.crn_ingest_wait.argtypes = [C.c_void_p]
        L.crn_ingest_dropped.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
        L.crn_ingest_packets_per_epoch.argtypes = [C.c_void_p, C.POINTER(C.c_int32)]
        L.crn_sense_reserve_host.argtypes = [C.c_void_p, C.c_int64, C.c_int32]
        L.crn_noise_floor_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_float), C.c_void_p]
        L.crn_sense_set_thresholds.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_int32, C.c_void_p]
        L.crn_sense_reserve_noise_floor.argtypes = [C.c_void_p]
        L.crn_sense_calibrate_thresholds.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.POINTER(C.c_float), C.c_void_p]
        L.crn_ingest_calibrate.argtypes = [C.c_void_p, C.c_int32, C.c_float]
        L.crn_ingest_noise_floor.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int32)]
        L.crn_sense_set_timing.argtypes = [C.c_void_p, C.c_int32]
        L.crn_sense_dealt_launches.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
        L.crn_sense_get_stats.argtypes = [C.c_void_p, C.POINTER(SenseStats)]
        L.crn_ingest_get_stats.argtypes = [C.c_void_p, C.POINTER(IngestStats)]
        L.crn_monitor_rows_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_float, C.c_int32,
                                              C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.crn_comm_unique_id.argtypes = [C.c_void_p]
        L.crn_comm_create.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int64, C.c_int32,
                                      C.POINTER(C.c_void_p)]
        L.crn_comm_local.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.POINTER(C.c_void_p)]
        L.crn_comm_allgather.argtypes = [C.c_void_p, C.c_int64, C.c_void_p]
        L.crn_comm_local_addr.argtypes = [C.c_void_p, C.c_int64, C.POINTER(C.c_void_p)]
        L.crn_comm_wait.argtypes = [C.c_void_p, C.c_int64, C.c_void_p]
        L.crn_comm_gathered.argtypes = [C.c_void_p, C.c_int64, C.POINTER(C.c_void_p)]
        L.crn_comm_finish.argtypes = [C.c_void_p, C.c_void_p]
        L.crn_comm_destroy.argtypes = [C.c_void_p]
        L.crn_comm_info.argtypes = [C.c_void_p, C.POINTER(CommInfo)]
        _lib = L
    return _lib


def has_sc16():
    """Does the loaded library carry the optional wire-format kernels and entry points (a `make SC16=1` build)?  (False, not an error,
    while the library has not been built yet: test modules ask at import time, before the fixture that builds it has run.)"""
    if _lib is None and not os.path.exists(LIB_PATH):
        return False
    return hasattr(lib(), "crn_sense_run_device_sc16")


def _need_sc16(what):
    if not has_sc16():
        raise CrnError(f"{what}: {LIB_PATH} was built without the optional wire-format kernels "
                       "(make -C csrc SC16=1 builds libcrnsense_sc16.so; select it with $CRN_SENSE_LIB)")


def build_info():
    """(built_hip, runtime_hip, compatible): HIP_VERSION the library was compiled with, the runtime's, and crn_build_info's verdict."""
    b, r = C.c_int32(), C.c_int32()
    rc = lib().crn_build_info(C.byref(b), C.byref(r))
    return b.value, r.value, rc == 0


def check(rc, what):
    if rc != 0:
        raise CrnError(f"{what} failed ({rc}): {lib().crn_last_error().decode()}")


def cfg_reference():
    c = Cfg()
    check(lib().crn_cfg_reference(C.byref(c)), "crn_cfg_reference")
    return c


def cfg_energy_scaled(fft_len, lam=4.0):
    c = Cfg()
    check(lib().crn_cfg_energy_scaled(C.byref(c), fft_len, lam), "crn_cfg_energy_scaled")
    return c


def cfg_reference_scaled(fft_len):
    c = Cfg()
    check(lib().crn_cfg_reference_scaled(C.byref(c), fft_len), "crn_cfg_reference_scaled")
    return c


def cfg_welch_scaled(fft_len, frames_per_epoch=8, lam=4.0):
    c = Cfg()
    check(lib().crn_cfg_welch_scaled(C.byref(c), fft_len, frames_per_epoch, lam), "crn_cfg_welch_scaled")
    return c


def save_ann(cfg, path):
    check(lib().crn_cfg_save_ann(C.byref(cfg), os.fsencode(path)), "crn_cfg_save_ann")


def load_ann(cfg, path):
    check(lib().crn_cfg_load_ann(C.byref(cfg), os.fsencode(path)), "crn_cfg_load_ann")
    return cfg


def cfg_welch(fft_len, frames_per_epoch, n_bands):
    c = Cfg()
    check(lib().crn_cfg_welch(C.byref(c), fft_len, frames_per_epoch, n_bands), "crn_cfg_welch")
    return c


def samples_per_epoch(cfg, L=None):
    """Dense epoch length in samples (the default epoch_stride)."""
    L = cfg.fft_len if L is None else L
    return cfg.frames_per_epoch * (L if cfg.hop == cfg.fft_len else cfg.hop)


def samples_needed(cfg, n_epochs, L=None):
    """Samples a dense buffer must hold for n_epochs (overlap adds a tail)."""
    L = cfg.fft_len if L is None else L
    tail = 0 if cfg.hop == cfg.fft_len else cfg.fft_len - cfg.hop
    return n_epochs * samples_per_epoch(cfg, L) + tail


class Sensor:
    """Owns one crn_handle."""

    def __init__(self, cfg):
        self.cfg = cfg
        self._h = C.c_void_p()
        # Ingest objects attached to this handle: closed before it (crn_sense_destroy refuses otherwise).  Weak references: a ring keeps
        # its sensor alive, not the other way round, so a dropped Ingest is destroyed by its refcount (launcher thread, pinned buffers)
        self._rings = weakref.WeakSet()
        check(lib().crn_sense_create(C.byref(cfg), C.byref(self._h)), "crn_sense_create")

    def close(self):
        if self._h:
            for ring in list(self._rings):
                ring.close()
            check(lib().crn_sense_destroy(self._h), "crn_sense_destroy")
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reserve_host(self, max_epochs, want_spectrum=False):
        check(lib().crn_sense_reserve_host(self._h, max_epochs, int(want_spectrum)), "crn_sense_reserve_host")

    def set_variant(self, v):
        check(lib().crn_sense_set_variant(self._h, v), "crn_sense_set_variant")

    def noise_floor(self, features_ptr, n_epochs, stream=0):
        """Median over epochs of each epoch's median band energy (features [n_epochs][n_bands] on the device)."""
        nf = C.c_float()
        check(lib().crn_noise_floor_device(self._h, features_ptr, n_epochs, C.byref(nf), stream), "crn_noise_floor_device")
        return nf.value

    def set_thresholds(self, thresh, stream=0):
        """Replace the per-band thresholds (ordered on `stream`); also updates self.cfg so that a checker sees the same values."""
        arr = (C.c_float * len(thresh))(*[float(t) for t in thresh])
        check(lib().crn_sense_set_thresholds(self._h, arr, len(thresh), stream), "crn_sense_set_thresholds")
        for b, t in enumerate(arr):
            self.cfg.thresh[b] = t

    def set_ann(self, w_ih, w_ho, threshold=0.8, stream=0):
        """Replace the network of a DECIDE_ANN handle (ordered on `stream`); also updates self.cfg."""
        import numpy as np
        a, b = np.ascontiguousarray(w_ih, np.float64), np.ascontiguousarray(w_ho, np.float64)
        assert a.shape == (5, 6) and b.shape == (6, 4)
        check(lib().crn_sense_set_ann(self._h, a.ctypes.data, b.ctypes.data, threshold, C.c_void_p(stream or None)), "crn_sense_set_ann")
        set_ann_weights(self.cfg, a, b, threshold)

    def set_bands(self, segs, n_bands, thresh=None):
        """Replace the band plan of the live handle: segs = [(lo, hi, band), ..]; thresh = n_bands floats or None (keep)."""
        arr = (BandSeg * len(segs))(*[BandSeg(int(lo), int(hi), int(b)) for lo, hi, b in segs])
        th = None if thresh is None else (C.c_float * n_bands)(*[float(t) for t in thresh])
        check(lib().crn_sense_set_bands(self._h, arr, len(segs), n_bands, th), "crn_sense_set_bands")
        self.cfg.n_segs, self.cfg.n_bands = len(segs), n_bands
        for i, sg in enumerate(arr):
            self.cfg.segs[i] = sg
        if th is not None:
            for b in range(n_bands):
                self.cfg.thresh[b] = th[b]

    def reserve_noise_floor(self):
        check(lib().crn_sense_reserve_noise_floor(self._h), "crn_sense_reserve_noise_floor")

    def calibrate_thresholds(self, features, lam, stream=0):
        """Upload a numpy [n_epochs][n_bands] float32 matrix, estimate the noise floor, set every threshold to lam x it; returns the
        estimate and updates self.cfg."""
        import numpy as np
        f = np.ascontiguousarray(features, np.float32)
        nf = C.c_float()
        check(lib().crn_sense_calibrate_thresholds(self._h, f.ctypes.data, f.shape[0], lam, C.byref(nf), C.c_void_p(stream or None)),
              "crn_sense_calibrate_thresholds")
        for b in range(self.cfg.n_bands):
            self.cfg.thresh[b] = float(np.float32(lam) * np.float32(nf.value))
        return nf.value

    def noise_floor_host(self, features):
        """crn_noise_floor_host on a numpy [n_epochs][n_bands] float32 matrix."""
        import numpy as np
        f = np.ascontiguousarray(features, np.float32)
        nf = C.c_float()
        check(lib().crn_noise_floor_host(self._h, f.ctypes.data, f.shape[0], C.byref(nf)), "crn_noise_floor_host")
        return nf.value

    def set_timing(self, on=True):
        check(lib().crn_sense_set_timing(self._h, 1 if on else 0), "crn_sense_set_timing")

    def stats(self):
        st = SenseStats()
        check(lib().crn_sense_get_stats(self._h, C.byref(st)), "crn_sense_get_stats")
        return {k: getattr(st, k) for k, _ in SenseStats._fields_}

    def dealt_launches(self):
        """Launches of this handle that ran the dealt-frame form of the kernel (small launches at 512 / 1024 points)."""
        n = C.c_int64()
        check(lib().crn_sense_dealt_launches(self._h, C.byref(n)), "crn_sense_dealt_launches")
        return n.value

    def kernel_info(self):
        name = C.create_string_buffer(256)
        thr, lds, epb = C.c_int32(), C.c_int32(), C.c_int32()
        check(lib().crn_sense_kernel_info(self._h, name, 256, C.byref(thr), C.byref(lds), C.byref(epb)),
              "crn_sense_kernel_info")
        return {"name": name.value.decode(), "threads_per_block": thr.value, "lds_bytes": lds.value,
                "epochs_per_block": epb.value}

    def run_device(self, iq_ptr, n_epochs, L, out_ptrs, stream=0, epoch_stride=0, sc16=False):
        """iq_ptr / out_ptrs: raw device addresses (ints); out_ptrs keys are Out fields.  sc16: iq_ptr holds the radio's wire
        format (int16 pairs, 4 bytes per complex sample)."""
        o = Out(**{k: (v or None) for k, v in out_ptrs.items()})
        if sc16:
            _need_sc16("run_device(sc16=True)")
        fn = lib().crn_sense_run_device_sc16 if sc16 else lib().crn_sense_run_device
        check(fn(self._h, iq_ptr, n_epochs, L, epoch_stride, C.byref(o), C.c_void_p(stream or None)),
              "crn_sense_run_device_sc16" if sc16 else "crn_sense_run_device")

    def set_wire_full_scale(self, full_scale):
        _need_sc16("set_wire_full_scale")
        check(lib().crn_sense_set_wire_full_scale(self._h, float(full_scale)), "crn_sense_set_wire_full_scale")

    def pack_sc16_device(self, iq_ptr, n_samples, out_ptr, stream=0):
        """complex floats -> int16 pairs on the device (n_samples complex samples)."""
        _need_sc16("pack_sc16_device")
        check(lib().crn_pack_sc16_device(self._h, iq_ptr, n_samples, out_ptr, C.c_void_p(stream or None)), "crn_pack_sc16_device")

    def run_host(self, iq, n_epochs, L=None, want_spectrum=False, epoch_stride=0):
        """iq: numpy float32 array of interleaved samples. Returns dict of numpy outputs."""
        import numpy as np
        cfg = self.cfg
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
        o = Out(**{k: v.ctypes.data for k, v in res.items()})
        check(lib().crn_sense_run_host(self._h, iq.ctypes.data, n_epochs, L, epoch_stride, C.byref(o)),
              "crn_sense_run_host")
        return res

    def synth_fill_device(self, iq_ptr, n_epochs, spe, seed, noise_power=1e-6, signal_rms=0.02,
                          tones=8, truth_ptr=0, stream=0):
        check(lib().crn_synth_fill_device(self._h, iq_ptr, n_epochs, spe, seed, noise_power, signal_rms,
                                          tones, C.c_void_p(truth_ptr or None), C.c_void_p(stream or None)),
              "crn_synth_fill_device")

    def synth_fill_device_ex(self, iq_ptr, n_epochs, spe, sc, truth_ptr=0, stream=0):
        """sc: SynthCfg (traffic model, signal kind, streams)."""
        check(lib().crn_synth_fill_device_ex(self._h, C.byref(sc), iq_ptr, n_epochs, spe,
                                             C.c_void_p(truth_ptr or None), C.c_void_p(stream or None)),
              "crn_synth_fill_device_ex")

    def fft_forward_device(self, in_ptr, n_frames, L, out_ptr, frame_stride=0, stream=0):
        """Unnormalised forward DFT of n_frames frames (device pointers)."""
        check(lib().crn_fft_forward_device(self._h, in_ptr, n_frames, L, frame_stride, out_ptr,
                                           C.c_void_p(stream or None)), "crn_fft_forward_device")

    def monitor_rows_device(self, spec_ptr, n_rows, kind, alpha, first, state_ptr, waterfall_ptr=0, average_ptr=0, stream=0):
        """fftshifted dB rows + IIR-averaged trace from rows of the `spectrum` output (device pointers)."""
        check(lib().crn_monitor_rows_device(self._h, spec_ptr, n_rows, kind, alpha, int(first), state_ptr,
                                            C.c_void_p(waterfall_ptr or None), C.c_void_p(average_ptr or None),
                                            C.c_void_p(stream or None)), "crn_monitor_rows_device")

    def ann_train_device(self, tc, feat_ptr, label_ptr, n, stream=0):
        """Fit the 4-5-3 network to device-resident features/labels; returns (w_ih, w_ho, loss)."""
        import numpy as np
        wih, who = np.zeros((5, 6), np.float64), np.zeros((6, 4), np.float64)
        loss = C.c_double()
        check(lib().crn_ann_train_device(self._h, C.byref(tc), feat_ptr, label_ptr, n, wih.ctypes.data,
                                         who.ctypes.data, C.byref(loss), C.c_void_p(stream or None)),
              "crn_ann_train_device")
        return wih, who, loss.value


def set_ann_weights(cfg, w_ih, w_ho, threshold=0.8):
    """Put trained weights into a crn_cfg and select the ANN + cascade decision."""
    for i in range(5):
        for j in range(6):
            cfg.ann_w_ih[i][j] = float(w_ih[i][j])
    for j in range(6):
        for k in range(4):
            cfg.ann_w_ho[j][k] = float(w_ho[j][k])
    cfg.ann_threshold = threshold
    cfg.decide = DECIDE_ANN
    return cfg


COMM_ID_BYTES = 128

_hip = None


def device_to_host(ptr, nbytes):
    """Blocking copy of device memory at raw address `ptr` into a bytes object (plumbing for checks)."""
    global _hip
    if _hip is None:
        _hip = C.CDLL("libamdhip64.so")
        _hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    buf = (C.c_uint8 * nbytes)()
    rc = _hip.hipMemcpy(buf, C.c_void_p(ptr), nbytes, 2)  # hipMemcpyDeviceToHost
    if rc != 0:
        raise CrnError(f"hipMemcpy D2H failed ({rc})")
    return bytes(buf)


def comm_unique_id():
    """bytes(128): call on rank 0, hand to the other ranks out of band."""
    buf = (C.c_uint8 * COMM_ID_BYTES)()
    check(lib().crn_comm_unique_id(buf), "crn_comm_unique_id")
    return bytes(buf)


class Comm:
    """The occupancy exchange over RCCL (crn_comm_* in include/crn_sense.h): `depth` slots of
    bytes_per_rank on `device`, all-gathers queued on a side stream."""

    def __init__(self, device, rank, world, unique_id, bytes_per_rank, depth=2):
        self.rank, self.world, self.bytes, self.depth = rank, world, bytes_per_rank, depth
        self._c = C.c_void_p()
        buf = (C.c_uint8 * COMM_ID_BYTES).from_buffer_copy(unique_id)
        check(lib().crn_comm_create(device, rank, world, buf, bytes_per_rank, depth, C.byref(self._c)), "crn_comm_create")

    def local(self, step, stream=0):
        p = C.c_void_p()
        check(lib().crn_comm_local(self._c, step, C.c_void_p(stream or None), C.byref(p)), "crn_comm_local")
        return p.value

    def local_addr(self, step):
        """The slot's address, without making any stream wait or releasing the slot."""
        p = C.c_void_p()
        check(lib().crn_comm_local_addr(self._c, step, C.byref(p)), "crn_comm_local_addr")
        return p.value

    def wait(self, step, stream=0):
        """Make `stream` wait for step's gather only; the slot stays in flight for local() / finish()."""
        check(lib().crn_comm_wait(self._c, step, C.c_void_p(stream or None)), "crn_comm_wait")

    def allgather(self, step, stream=0):
        check(lib().crn_comm_allgather(self._c, step, C.c_void_p(stream or None)), "crn_comm_allgather")

    def gathered(self, step):
        p = C.c_void_p()
        check(lib().crn_comm_gathered(self._c, step, C.byref(p)), "crn_comm_gathered")
        return p.value

    def finish(self, stream=0):
        check(lib().crn_comm_finish(self._c, C.c_void_p(stream or None)), "crn_comm_finish")

    def info(self):
        """What RCCL says the communicator is (ncclCommCount / UserRank / CuDevice / GetVersion), as a dict."""
        ci = CommInfo()
        check(lib().crn_comm_info(self._c, C.byref(ci)), "crn_comm_info")
        return {"nranks": ci.nranks, "rank": ci.rank, "rccl_device": ci.rccl_device, "rccl_version": ci.rccl_version,
                "device": ci.device, "depth": ci.depth, "bytes_per_rank": ci.bytes_per_rank, "gathers": ci.gathers,
                "library": ci.library.decode(errors="replace"), "pci_bus_id": ci.pci_bus_id.decode(errors="replace")}

    def close(self):
        if self._c:
            lib().crn_comm_destroy(self._c)
            self._c = C.c_void_p()


class Ingest:
    """Packet ingest ring over a Sensor (crn_ingest_* in include/crn_sense.h)."""

    def __init__(self, sensor, n_streams, samples_per_packet, epochs_per_batch, sc16=False):
        """sc16: packets are int16 pairs (the radio's wire format) instead of complex floats."""
        self.sensor = sensor
        self._g = C.c_void_p()
        if sc16:
            _need_sc16("Ingest(sc16=True)")
        self._push = lib().crn_ingest_push_sc16 if sc16 else lib().crn_ingest_push
        create = lib().crn_ingest_create_sc16 if sc16 else lib().crn_ingest_create
        check(create(sensor._h, n_streams, samples_per_packet, epochs_per_batch, C.byref(self._g)), "crn_ingest_create")
        sensor._rings.add(self)

    def calibrate(self, n_epochs, lam):
        """Post a noise-floor calibration over the next n_epochs epochs (carried out by the ring's launcher thread)."""
        check(lib().crn_ingest_calibrate(self._g, n_epochs, lam), "crn_ingest_calibrate")

    def noise_floor(self):
        """(estimate in force, calibration collecting?)"""
        nf, busy = C.c_float(), C.c_int32()
        check(lib().crn_ingest_noise_floor(self._g, C.byref(nf), C.byref(busy)), "crn_ingest_noise_floor")
        return nf.value, bool(busy.value)

    def push(self, stream, packet, block=True):
        """block=True: on CRN_ERR_BUSY wait for a free buffer and push again (tests, tools);
        block=False: returns False when the packet was refused (what an engine's execute() does)."""
        rc = self._push(self._g, stream, packet.ctypes.data)
        while rc == CRN_ERR_BUSY and block:
            check(lib().crn_ingest_wait(self._g), "crn_ingest_wait")
            rc = self._push(self._g, stream, packet.ctypes.data)
        if rc == CRN_ERR_BUSY:
            return False
        check(rc, "crn_ingest_push")
        return True

    def set_packet_len(self, L):
        check(lib().crn_ingest_set_packet_len(self._g, L), "crn_ingest_set_packet_len")

    def stats(self):
        st = IngestStats()
        check(lib().crn_ingest_get_stats(self._g, C.byref(st)), "crn_ingest_get_stats")
        return {k: getattr(st, k) for k, _ in IngestStats._fields_}

    def packets_per_epoch(self):
        n = C.c_int32()
        check(lib().crn_ingest_packets_per_epoch(self._g, C.byref(n)), "crn_ingest_packets_per_epoch")
        return n.value

    def dropped(self):
        n = C.c_int64()
        check(lib().crn_ingest_dropped(self._g, C.byref(n)), "crn_ingest_dropped")
        return n.value

    def flush(self):
        check(lib().crn_ingest_flush(self._g), "crn_ingest_flush")

    def drain(self):
        check(lib().crn_ingest_drain(self._g), "crn_ingest_drain")

    def poll(self, max_results=64):
        arr = (EpochResult * max_results)()
        n = C.c_int32()
        check(lib().crn_ingest_poll(self._g, arr, max_results, C.byref(n)), "crn_ingest_poll")
        return [arr[i] for i in range(n.value)]

    def close(self):
        if self._g:
            lib().crn_ingest_destroy(self._g)
            self._g = C.c_void_p()
            self.sensor._rings.discard(self)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
