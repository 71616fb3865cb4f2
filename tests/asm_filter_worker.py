"""Child process of tests/test_asm_filter.py: run every kernel family of the library $CRN_SENSE_LIB names (libcrnsense.so /
libcrnsense_sc16.so: built through csrc/strip_asm_nops.py; libcrnsense_plain.so: plain hipcc) on generated batches and print one JSON
object {case: {output: sha256}} — the parent asserts that the filtered and the plain build give the same BYTES.

    python tests/asm_filter_worker.py [--big-gib 2.0]

Inputs are made by torch (a seeded generator on the device, the same in every process), not by the library under test: noise plus a
few strong tones, so that accumulators and band sums are far from any value that would hide a different bit.
"""
import argparse
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(os.path.dirname(HERE), "cognitive-radio-network_amd"), HERE]
import numpy as np  # noqa: E402
import torch  # noqa: E402

import crnsense as cs  # noqa: E402


def sha(t):
    return hashlib.sha256(t.detach().cpu().contiguous().numpy().tobytes()).hexdigest()


def make_iq(n_samples, seed, dev):
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    iq = torch.randn(n_samples * 2, generator=g, device=dev, dtype=torch.float32) * 7.07e-4
    k = torch.arange(n_samples, device=dev, dtype=torch.float32)
    for f, a in ((0.01171875, 0.02), (0.1513671875, 0.013), (-0.3, 0.004)):   # strong carriers, on and off the FFT grids
        ph = 2 * np.pi * f * k
        iq[0::2] += a * torch.cos(ph)
        iq[1::2] += a * torch.sin(ph)
    return iq


def run_case(name, cfg, dev, n_epochs, L=None, variant=0, deal=400, want_spectrum=False, gap=0, sc16=False, seed=1):
    L = cfg.fft_len if L is None else L
    spe = cs.samples_per_epoch(cfg, L)
    stride = spe + gap if gap else 0
    need = (n_epochs - 1) * (stride or spe) + cs.samples_needed(cfg, 1, L)
    iq = make_iq(need, seed, dev)
    src = iq
    if sc16:
        src = torch.clamp(torch.round(iq * (32768.0 * 16.0)), -32768, 32767).to(torch.int16)
    s = cs.Sensor(cfg)
    s.set_variant(variant)
    s.set_variant(deal)
    out = {"features": torch.full((n_epochs, cfg.n_bands), -1.0, device=dev),
           "ann_out": torch.full((n_epochs, 3), -1.0, dtype=torch.float64, device=dev),
           "decision": torch.full((n_epochs,), -7, dtype=torch.int32, device=dev),
           "occupancy": torch.full((n_epochs, cfg.n_bands), 9, dtype=torch.uint8, device=dev)}
    if want_spectrum:
        out["spectrum"] = torch.full((n_epochs, cfg.fft_len), -1.0, device=dev)
    ptrs = {k: v.data_ptr() for k, v in out.items()}
    ptrs.setdefault("spectrum", 0)
    s.run_device(src.data_ptr(), n_epochs, L, ptrs, epoch_stride=stride, sc16=sc16)
    torch.cuda.synchronize()
    info = {"kernel": s.kernel_info()["name"], "dealt_launches": s.dealt_launches(), "input_bytes": int(src.numel() * src.element_size())}
    s.close()
    res = {k: sha(v) for k, v in out.items()}
    res["_info"] = info
    del iq, src, out
    torch.cuda.empty_cache()
    return res


def cases(big_gib):
    """(name, kwargs): every family of sense_kernel / sense_kernel_dealt the dispatch can reach, the big streaming ones on >= big_gib."""
    def epochs_for(cfg, gib, L=None):
        return max(1, int(gib * 2 ** 30) // (cs.samples_per_epoch(cfg, cfg.fft_len if L is None else L) * 8))
    for n in (512, 1024, 2048, 4096):
        c = cs.cfg_energy_scaled(n, 4.0)
        yield f"energy{n} reference plan (pruned, register close) {big_gib} GiB", dict(cfg=c, n_epochs=epochs_for(c, big_gib))
        yield f"energy{n} L=364 short packets", dict(cfg=c, n_epochs=4099, L=364)
        yield f"energy{n} + spectrum (LDS close)", dict(cfg=c, n_epochs=517, want_spectrum=True)
        m = cs.cfg_reference_scaled(n)
        yield f"|X| mode {n} + network, whole frames {big_gib / 4} GiB", dict(cfg=m, n_epochs=epochs_for(m, big_gib / 4))
        yield f"|X| mode {n} + network, L=364", dict(cfg=m, n_epochs=4099, L=364)
        yield f"|X| mode {n} + spectrum", dict(cfg=m, n_epochs=131, want_spectrum=True)
        o = cs.cfg_energy_scaled(n, 4.0)                 # a band outside the reference plan's rows: register close without pruning
        o.segs[2].lo, o.segs[2].hi = n // 4 + 3, n // 4 + 41
        yield f"energy{n} other small plan (register close, unpruned)", dict(cfg=o, n_epochs=2051)
        yield f"energy{n} other small plan, L=100 (LDS walk)", dict(cfg=o, n_epochs=515, L=100)
        b = cs.cfg_energy_scaled(n, 4.0)                 # 16 equal bands: the LDS walk
        b.n_bands, b.n_segs, b.ref_band = 16, 16, -1
        for i in range(16):
            b.segs[i].lo, b.segs[i].hi, b.segs[i].band = i * (n // 16), (i + 1) * (n // 16), i
            b.thresh[i] = 1e-3
        yield f"energy{n} 16 bands (LDS walk)", dict(cfg=b, n_epochs=1031)
        w = cs.cfg_welch(n, 8, 64)
        for i in range(64):
            w.thresh[i] = 4.0 * (n / 64) * n * 1e-6 * 0.375
        yield f"welch{n} stream (Hann in pass 1) {big_gib / 2} GiB", dict(cfg=w, n_epochs=epochs_for(w, big_gib / 2))
        yield f"welch{n} epochs with gaps (one epoch group per workgroup)", dict(cfg=w, n_epochs=1027, gap=333)
        yield f"welch{n} + spectrum", dict(cfg=w, n_epochs=67, want_spectrum=True)
        t = cs.cfg_energy_scaled(n, 4.0)
        t.window = cs.WINDOW_BLACKMAN_HARRIS
        yield f"table window {n} energy", dict(cfg=t, n_epochs=1031)
        t2 = cs.cfg_reference_scaled(n)
        t2.window = cs.WINDOW_HANN
        yield f"table window {n} |X| L=364", dict(cfg=t2, n_epochs=515, L=364)
    c = cs.cfg_energy_scaled(4096, 4.0)
    yield "energy4096 variant 2 (unpruned)", dict(cfg=c, n_epochs=epochs_for(c, big_gib), variant=2)
    for n in (512, 1024):                                # launches of a few epochs: the dealt-frame kernels
        for nm, c in (("reference", cs.cfg_reference_scaled(n)), ("energy", cs.cfg_energy_scaled(n, 4.0)), ("welch", cs.cfg_welch(n, 8, 64))):
            if nm == "welch":
                for i in range(64):
                    c.thresh[i] = 1e-3
            yield f"dealt {nm} {n}, 1 epoch", dict(cfg=c, n_epochs=1, deal=402)
            yield f"dealt {nm} {n}, 200 epochs", dict(cfg=c, n_epochs=200, deal=402)
        yield f"dealt reference {n} L=364 + spectrum", dict(cfg=cs.cfg_reference_scaled(n), n_epochs=9, L=364, deal=402, want_spectrum=True)
    if cs.has_sc16():
        for n in (512, 4096):
            c = cs.cfg_energy_scaled(n, 4.0)
            yield f"sc16 energy{n} {big_gib / 2} GiB", dict(cfg=c, n_epochs=epochs_for(c, big_gib), sc16=True)   # (4 B per sample)
            yield f"sc16 |X| mode {n} L=364", dict(cfg=cs.cfg_reference_scaled(n), n_epochs=4099, L=364, sc16=True)
            w = cs.cfg_welch(n, 8, 64)
            for i in range(64):
                w.thresh[i] = 1e-3
            yield f"sc16 welch{n}", dict(cfg=w, n_epochs=2051, sc16=True)
        yield "sc16 dealt reference 512 L=364", dict(cfg=cs.cfg_reference(), n_epochs=3, L=364, deal=402, sc16=True)


def fft_case(dev):
    """crn_fft_forward_device: the transform on its own (the same butterflies)."""
    import ctypes as C
    out = {}
    for n in (512, 1024, 2048, 4096):
        cfg = cs.cfg_energy_scaled(n, 4.0)
        s = cs.Sensor(cfg)
        x = make_iq(257 * n, 5, dev)
        y = torch.empty(257 * n * 2, device=dev)
        cs.check(cs.lib().crn_fft_forward_device(s._h, x.data_ptr(), 257, n, 0, y.data_ptr(), C.c_void_p(None)), "crn_fft_forward_device")
        torch.cuda.synchronize()
        out[f"fft{n}"] = sha(y)
        s.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--big-gib", type=float, default=2.0)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    res = {"_library": os.path.basename(cs.LIB_PATH)}
    for i, (name, kw) in enumerate(cases(a.big_gib)):
        res[name] = run_case(name, dev=dev, seed=100 + i, **kw)
    res["forward FFT"] = fft_case(dev)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
