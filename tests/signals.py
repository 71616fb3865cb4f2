"""Seeded synthetic IQ for parity tests (SURVEY.md §8(d) recipe, Appendix C amplitudes).

fs = 13 MHz, complex fp32.  AWGN with E|x|^2 = 1e-6 everywhere; an occupied band carries 8
random-phase on-grid tones spread over its bins, total RMS amplitude 0.02.  With the reference
parameters (N = 512, K = 10) this puts idle features near 0.04 (NF) / 0.4 (channels) and an
occupied channel near 8e2, where the shipped ANN weights are decisive.
"""
import numpy as np


def band_bins(cfg, band):
    bins = []
    for s in range(cfg.n_segs):
        g = cfg.segs[s]
        if g.band == band:
            bins.extend(range(g.lo, g.hi))
    return np.array(bins, dtype=np.int64)


def make_epochs(cfg, n_epochs, seed, L=None, picks=None, noise_power=1e-6, signal_rms=0.02, tones=8):
    """Returns (iq float32 [samples*2], picks int array [n_epochs] with 0 = idle, b = band b driven)."""
    N, K = cfg.fft_len, cfg.frames_per_epoch
    L = N if L is None else L
    rng = np.random.default_rng(seed)
    overlap = cfg.hop != N
    spe = K * (cfg.hop if overlap else L)
    total = n_epochs * spe + (N - cfg.hop if overlap else 0)
    sig = np.sqrt(noise_power / 2)
    x = (rng.normal(0, sig, total) + 1j * rng.normal(0, sig, total)).astype(np.complex64)
    if picks is None:
        hi = min(3, cfg.n_bands - 1)
        picks = rng.integers(0, hi + 1, n_epochs)
    picks = np.asarray(picks)
    n = np.arange(spe + (N - cfg.hop if overlap else 0))
    for e in range(n_epochs):
        b = int(picks[e])
        if b <= 0:
            continue
        bins = band_bins(cfg, b)
        nt = min(tones, bins.size)
        ks = bins[((2 * np.arange(nt) + 1) * bins.size) // (2 * nt)]
        amp = signal_rms / np.sqrt(nt)
        ph = rng.uniform(0, 2 * np.pi, nt)
        # phase continuous in the frame-local index so tones stay on-grid for every frame
        nn = n if overlap else (n % L)
        s = np.zeros(n.size, np.complex128)
        for k, p in zip(ks, ph):
            s += amp * np.exp(1j * (2 * np.pi * k * nn / N + p))
        seg = slice(e * spe, e * spe + n.size)
        x[seg] = (x[seg].astype(np.complex128) + s[: x[seg].size]).astype(np.complex64)
    return x.view(np.float32).copy(), picks


def spectrum_f64(cfg, iq, n_epochs, L=None):
    """float64 ground truth of the K-frame per-bin average (numpy pocketfft on complex128)."""
    N, K = cfg.fft_len, cfg.frames_per_epoch
    L = N if L is None else L
    x = np.asarray(iq, np.float32).view(np.complex64).astype(np.complex128)
    overlap = cfg.hop != N
    fs = cfg.hop if overlap else L
    spe = K * fs
    w = np.ones(N)
    if cfg.window == 1:
        w = (0.5 - 0.5 * np.cos(2 * np.pi * np.arange(N) / N)).astype(np.float32).astype(np.float64)
    elif cfg.window == 2:
        th = 2 * np.pi * np.arange(N) / (N - 1)
        w = (0.35875 - 0.48829 * np.cos(th) + 0.14128 * np.cos(2 * th) - 0.01168 * np.cos(3 * th)).astype(np.float32).astype(np.float64)
    out = np.zeros((n_epochs, N))
    for e in range(n_epochs):
        for f in range(K):
            fr = np.zeros(N, np.complex128)
            st = e * spe + f * fs
            fr[:L] = x[st:st + L]
            X = np.fft.fft(fr * w)
            out[e] += np.abs(X) / K if cfg.mode == 0 else (X.real ** 2 + X.imag ** 2) / K
    return out
