"""Independent float64 restatement of the sensing path, end to end.  TEST INFRASTRUCTURE.

Why this file exists: the C oracle (oracle/crn_oracle.c) and the HIP path are both checked against
tests/golden/.  Those fixtures used to be written by the C oracle itself — a regression pin, not
evidence.  This module is a second, separately written statement of the same mathematics that
shares NO code with oracle/ (pure numpy, no ctypes, no crnsense / oracle_py import; constants typed
from SURVEY.md Appendix A, which tabulates CE_Predictive_Node.cpp:78-120,173-191), computes
everything in float64, and is what tests/golden/make_golden.py now records.  The C oracle and the
GPU then have to agree with numbers neither of them produced.

Semantics followed (reference: cognitive_engines/CE_Predictive_Node/CE_Predictive_Node.cpp):
  staging   :148-149   x[0:L] = packet, x[L:N] = 0
  transform :150       X = unnormalised forward DFT of x           (numpy pocketfft, complex128)
  mean      :152-154   a[k] += |X[k]| / K                          (mode "mag")
                       P[k] += |X[k]|^2 / K                        (mode "energy": build-side, crn_sense.h)
  band sums :173-191   M_b = sum of a[k] over the band's runs
  features  :194-197   F_b = M_b * M_b (mag) / F_b = M_b (energy)
  network   :200-235   H_j = s(W_IH[0][j] + sum_i F_i W_IH[i][j]),  O_k = s(W_HO[0][k] + sum_j H_j W_HO[j][k])
  cascade   :245-261   first O_k >= 0.8 wins, else none
Float64 throughout, so results differ from the reference's fp32 arithmetic by rounding only: the
tolerances that the fp32 implementations must meet against these values are stated in the tests.
"""
from dataclasses import dataclass, field

import numpy as np

# ---- constants, typed from SURVEY.md Appendix A -------------------------------------------------
# Band runs [lo, hi) at N = 512 in the reference's order of summation; band 0 = NF, 1..3 = CH1..CH3.
REF_RUNS_512 = {
    0: ((300, 310),),            # NF   = sum a[300..309]
    1: ((0, 16), (496, 511)),    # CH1  = a[0..15] then a[496..510]  (bin 511 is not part of it)
    2: ((55, 85),),              # CH2  = a[55..84]
    3: ((189, 222),),            # CH3  = a[189..221]
}
# W_IH[i][j]: i = 0 bias, 1 NF, 2 CH1, 3 CH2, 4 CH3; j = 1..5 (column 0 unused)
W_IH = np.array([
    [0.0, -0.188208, -0.170684, -0.024726, 0.001448, 0.015983],
    [0.0, -0.106634, -0.415470, 0.309261, 0.159974, 0.212781],
    [0.0, 0.005650, 0.741944, 0.006133, -0.620100, 0.669892],
    [0.0, -0.057578, 0.621154, -0.048268, -0.249186, 0.734475],
    [0.0, 0.092680, 0.809336, -0.010821, -0.546496, 0.609384],
])
# W_HO[j][k]: j = 0 bias, 1..5 hidden; k = 1..3 (column 0 unused)
W_HO = np.array([
    [0.0, -7.033320, 2.726400, -2.590206],
    [0.0, 10.857465, -18.452471, 15.609466],
    [0.0, -6.848443, 2.053071, -2.929559],
    [0.0, 17.053079, -13.375309, -15.703407],
    [0.0, 0.087664, -0.269499, 0.407028],
    [0.0, -6.552455, 2.655529, -2.552555],
])
ANN_THRESHOLD = 0.8
TX_FREQ = {0: None, 1: 835e6, 2: 833e6, 3: 835e6}   # .cpp:247,252,257; "ALL BUSY" tunes nothing


@dataclass
class Plan:
    """What one sensing configuration computes (the independent twin of crn_cfg)."""
    n: int
    k: int = 10
    hop: int = 0                      # 0 = n (disjoint frames)
    mode: str = "mag"                 # "mag" | "energy"
    window: str = "rect"              # "rect" | "hann" | "bh"
    runs: dict = field(default_factory=dict)   # band -> ((lo, hi), ...)
    decide: str = "ann"               # "ann" | "threshold" | "none"
    thresh: tuple = ()
    ref_band: int = -1
    w_ih: np.ndarray = None
    w_ho: np.ndarray = None
    ann_threshold: float = ANN_THRESHOLD

    @property
    def n_bands(self):
        return len(self.runs)


def plan_reference():
    """The reference engine's own parameters (CE_Predictive_Node.hpp:31-32, .cpp:173-191)."""
    return Plan(n=512, k=10, mode="mag", runs=dict(REF_RUNS_512), decide="ann", w_ih=W_IH, w_ho=W_HO)


def plan_energy_scaled(n, lam=4.0):
    """crn_sense.h `crn_cfg_energy_scaled`: the reference's runs scaled by n/512, energy mode,
    threshold relative to the noise-floor band: thr_b = lam * bins_b / bins_NF (fp32 values)."""
    s = n // 512
    runs = {b: tuple((lo * s, hi * s) for lo, hi in rr) for b, rr in REF_RUNS_512.items()}
    bins = {b: sum(hi - lo for lo, hi in rr) for b, rr in runs.items()}
    thr = [np.inf] + [float(np.float32(lam) * np.float32(bins[b]) / np.float32(bins[0])) for b in (1, 2, 3)]
    return Plan(n=n, k=10, mode="energy", runs=runs, decide="threshold", thresh=tuple(thr), ref_band=0,
                w_ih=W_IH, w_ho=W_HO)


def plan_welch(n, k, n_bands, thresh):
    """crn_sense.h `crn_cfg_welch`: Hann, hop n/2, n_bands equal contiguous bands, absolute thresholds."""
    w = n // n_bands
    runs = {b: ((b * w, (b + 1) * w),) for b in range(n_bands)}
    return Plan(n=n, k=k, hop=n // 2, mode="energy", window="hann", runs=runs, decide="threshold",
                thresh=tuple(float(t) for t in thresh), ref_band=-1, w_ih=W_IH, w_ho=W_HO)


def window(kind, n):
    """Window values as the fp32 tables the implementations multiply by (crn_sense.h crn_window)."""
    i = np.arange(n, dtype=np.float64)
    if kind == "rect":
        return np.ones(n)
    if kind == "hann":        # periodic Hann
        w = 0.5 - 0.5 * np.cos(2 * np.pi * i / n)
    elif kind == "bh":        # 4-term Blackman-Harris over n-1 (spectrum_analyzer.py:262-275)
        x = 2 * np.pi * i / (n - 1)
        w = 0.35875 - 0.48829 * np.cos(x) + 0.14128 * np.cos(2 * x) - 0.01168 * np.cos(3 * x)
    else:
        raise ValueError(kind)
    return w.astype(np.float32).astype(np.float64)


def spectrum(plan, iq, n_epochs, L=None, epoch_stride=0):
    """K-frame per-bin average, float64 [n_epochs, n]: fft_avg[] of CE_Predictive_Node.hpp:51."""
    n, k = plan.n, plan.k
    hop = plan.hop or n
    L = n if L is None else L
    x = np.asarray(iq, dtype=np.float32).view(np.complex64).astype(np.complex128)
    fstride = L if hop == n else hop
    stride = epoch_stride if epoch_stride > 0 else k * fstride
    w = window(plan.window, n)
    out = np.zeros((n_epochs, n))
    for e in range(n_epochs):
        for f in range(k):
            buf = np.zeros(n, dtype=np.complex128)
            st = e * stride + f * fstride
            buf[:L] = x[st:st + L]
            X = np.fft.fft(buf * w)
            out[e] += (np.abs(X) if plan.mode == "mag" else X.real ** 2 + X.imag ** 2) / k
    return out


def band_sums(plan, spec):
    m = np.zeros((spec.shape[0], plan.n_bands))
    for b, rr in plan.runs.items():
        for lo, hi in rr:
            m[:, b] += spec[:, lo:hi].sum(axis=1)
    return m


def features(plan, m):
    return m * m if plan.mode == "mag" else m


def sigmoid(s):
    return 1.0 / (1.0 + np.exp(-s))


def ann(feat4, w_ih=W_IH, w_ho=W_HO):
    """feat4 [.., 4] = {NF, CH1, CH2, CH3} -> outputs [.., 3] (Output[1..3])."""
    f = np.asarray(feat4, dtype=np.float64)
    with np.errstate(over="ignore"):   # exp(-s) -> inf for very negative s is benign: 1 / (1 + inf) = 0
        h = sigmoid(w_ih[0, 1:] + f @ w_ih[1:, 1:])
        return sigmoid(w_ho[0, 1:] + h @ w_ho[1:, 1:])


def cascade(out3, thr=ANN_THRESHOLD):
    o = np.atleast_2d(out3)
    d = np.zeros(o.shape[0], dtype=np.int32)
    for i, row in enumerate(o):
        for kk in range(3):
            if row[kk] >= thr:
                d[i] = kk + 1
                break
    return d


def run(plan, iq, n_epochs, L=None, epoch_stride=0):
    """Everything one launch produces, in float64.  `margin` is how far each epoch sits from a
    decision boundary (min |O - 0.8| or min |F / thr - 1|): fixtures assert it is large."""
    spec = spectrum(plan, iq, n_epochs, L, epoch_stride)
    feat = features(plan, band_sums(plan, spec))
    res = {"spectrum": spec, "features": feat, "ann_out": np.zeros((n_epochs, 3)),
           "decision": np.zeros(n_epochs, np.int32), "occupancy": np.zeros((n_epochs, plan.n_bands), np.uint8),
           "margin": np.full(n_epochs, np.inf)}
    if plan.decide == "ann":
        o = ann(feat, plan.w_ih, plan.w_ho)
        d = cascade(o, plan.ann_threshold)
        res["ann_out"], res["decision"] = o, d
        for e in range(n_epochs):
            if d[e] > 0:
                res["occupancy"][e, d[e]] = 1
        res["margin"] = np.abs(o - plan.ann_threshold).min(axis=1)
    elif plan.decide == "threshold":
        thr = np.asarray(plan.thresh, dtype=np.float64)[None, :]
        ref = feat[:, plan.ref_band:plan.ref_band + 1] if plan.ref_band >= 0 else 1.0
        # an infinite threshold ("never occupied": the noise-floor band itself) stays infinite whatever it is relative to —
        # inf x 0 (an empty reference band) would be NaN, and a NaN limit drops the row out of every margin comparison
        with np.errstate(invalid="ignore"):
            lim = np.where(np.isinf(thr), np.inf, thr * ref)
        occ = feat > lim
        res["occupancy"] = occ.astype(np.uint8)
        res["decision"] = occ.sum(axis=1).astype(np.int32)
        with np.errstate(divide="ignore", invalid="ignore"):
            rel = np.where(np.isfinite(lim) & (lim > 0), np.abs(feat / lim - 1.0), np.inf)   # lim = inf or 0: decided whatever the rounding
        res["margin"] = rel.min(axis=1)
    return res


def crossing(channel, nf=0.04, idle=0.39, thr=ANN_THRESHOLD):
    """Feature value at which a single occupied channel's output crosses `thr`, the other two
    channels idle (~231 / 25 / 45 for CH1 / CH2 / CH3), by bisection."""
    def out(v):
        f = np.array([nf, idle, idle, idle])
        f[channel] = v
        return ann(f)[channel - 1]
    lo, hi = idle, 1e4
    assert out(lo) < thr <= out(hi)
    for _ in range(200):
        mid = 0.5 * (lo + hi)
        lo, hi = (mid, hi) if out(mid) < thr else (lo, mid)
    return hi
