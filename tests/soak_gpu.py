#!/usr/bin/env python3
"""Soak: many random configurations through the HIP path against the oracle (not part of the
test suite; used to hunt rare codegen / aliasing bugs with spare GPU minutes)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "cognitive-radio-network_amd"), os.path.join(ROOT, "tests")]  # run as: python tests/soak_gpu.py [n]
import torch
import os as _os
_os.environ.setdefault("CRN_SENSE_AB", "1")   # measurement variants: libcrnsense_ab.so
import crnsense as cs, oracle_py as orc, signals

n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 200
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0   # python tests/soak_gpu.py <n> <first seed>: seeds first .. first + n - 1
bad = 0
for seed in range(first, first + n_seeds):
    rng = np.random.default_rng(50000 + seed)
    n = int(rng.choice([512, 1024, 2048, 4096]))
    ref_plan = rng.random() < 0.35
    cfg = cs.cfg_energy_scaled(n, 4.0)
    cfg.mode = int(rng.integers(0, 2))
    cfg.frames_per_epoch = int(rng.integers(1, 14))
    cfg.window = int(rng.choice([0, 0, 1, 2]))
    L = n if (cfg.window != 0 or rng.random() < 0.6) else int(rng.integers(1, n + 1))
    if not ref_plan:
        nb = int(rng.integers(1, 20))
        edges = np.sort(rng.choice(np.arange(1, n), size=2 * nb, replace=False))
        cfg.n_bands, cfg.n_segs, cfg.ref_band = nb, nb, -1
        for b in range(nb):
            cfg.segs[b].lo, cfg.segs[b].hi, cfg.segs[b].band = int(edges[2 * b]), int(edges[2 * b + 1]), b
            cfg.thresh[b] = 1e-3
    aligned = rng.random() < 0.15   # the Welch scan's plan: equal contiguous bands (N = 4096 takes the DPP band-sum close)
    if aligned:
        nb = int(rng.choice([16, 32, 64]))
        cfg.n_bands, cfg.n_segs, cfg.ref_band = nb, nb, int(rng.choice([-1, -1, 3]))
        for b in range(nb):
            cfg.segs[b].lo, cfg.segs[b].hi, cfg.segs[b].band = b * (n // nb), (b + 1) * (n // nb), b
            cfg.thresh[b] = 1e-3 if cfg.ref_band < 0 else 1.0
        cfg.decide = int(rng.choice([1, 1, 2]))
        if rng.random() < 0.7:
            cfg.window, cfg.mode, L = 1, 1, n
    variant = int(rng.choice([0, 0, 0, 2, 7])) if n == 4096 else 0
    if cfg.window == 1 and cfg.mode == 1 and L == n and rng.random() < 0.5:
        variant = int(rng.choice([19, 20, 21, 22]))   # A/B set of the windowed kernel
    if not cs.LIB_PATH.endswith("libcrnsense_ab.so") and variant not in (0, 2):
        variant = 0                                   # (a library without the measurement forms: $CRN_SENSE_LIB)
    want_spec = bool(rng.random() < 0.5)
    n_epochs = int(rng.integers(1, 40))
    # Welch (hop N/2, windowed, whole frames) for a fifth of the windowed cases
    welch = cfg.window != 0 and L == n and rng.random() < 0.4
    if welch:
        cfg.hop = n // 2
    # launch geometry: epoch groups per workgroup, single-group workgroups at the end, epoch stride
    epw = int(rng.choice([0, 0, 1, 2, 3, 4, 7]))
    tail = int(rng.choice([-1, -1, 0, 1]))
    spe = cs.samples_per_epoch(cfg, L)
    stride = 0
    if rng.random() < 0.25:
        stride = spe + int(rng.integers(0, 2 * n))  # gaps (dense epochs otherwise)
    need = (n_epochs - 1) * (stride or spe) + cs.samples_needed(cfg, 1, L)
    # small launches at 512 / 1024 points run the dealt-frame form of the kernel by themselves: a third of the cases leave that choice
    # alone, a third switch it off (the streaming form), a third force it
    deal = int(rng.choice([400, 401, 402]))
    iq = rng.normal(0, 1e-3, need * 2).astype(np.float32)
    s = cs.Sensor(cfg)
    s.set_variant(variant)
    s.set_variant(deal)
    if epw:
        s.set_variant(100 + epw)
    if tail >= 0:
        s.set_variant(200 + tail)
    got = s.run_host(iq, n_epochs, L=L, want_spectrum=want_spec, epoch_stride=stride)
    s.close()
    want = orc.run(cfg, iq, n_epochs, L=L, want_spectrum=want_spec, epoch_stride=stride)
    why = []
    _p64 = []
    def p64():
        """float64 per-bin K-frame means, epoch by epoch (each epoch on its own: dense or with gaps, disjoint or overlapped frames)"""
        if not _p64:
            sp = stride or spe
            extent = cs.samples_needed(cfg, 1, L)   # K L samples, or (K - 1) hop + N for overlapped frames
            _p64.append(np.concatenate([signals.spectrum_f64(cfg, iq[2 * e * sp: 2 * (e * sp + extent)], 1, L=L) for e in range(n_epochs)]))
        return _p64[0]
    ok = np.allclose(got["features"], want["features"], rtol=3e-5, atol=0)
    if not ok:
        # a band of one or two weak bins in a short epoch: the fp32 radix-2 restatement itself can sit 5e-5 from float64 there (seed
        # 308811: one bin, K = 1, oracle 5.1e-5 off, GPU 5.2e-6).  Decide against float64: the GPU must be within the bar of the
        # truth wherever the oracle disagrees with it, and never further from it than the oracle is
        # (a single frame can leave a bin at 1 % of the mean, where fp32 has no 1e-5: seeds 324579 / 358740, K = 1, GPU 2.3e-5 / 3.0e-5
        # from float64 and the oracle 3.8e-5 / 4.4e-5 — so the bar is taken against the per-bin floor of tests/parity_policy.py summed
        # over the band: 1e-1 x the mean bin below four frames, 1e-2 x from four on)
        f64 = np.zeros_like(want["features"], dtype=np.float64)
        nbins = np.zeros(cfg.n_bands)
        for i in range(cfg.n_segs):
            f64[:, cfg.segs[i].band] += p64()[:, cfg.segs[i].lo:cfg.segs[i].hi].sum(axis=1)
            nbins[cfg.segs[i].band] += cfg.segs[i].hi - cfg.segs[i].lo
        floor = (1e-2 if cfg.frames_per_epoch >= 4 else 1e-1) * p64().mean(axis=1, keepdims=True) * nbins[None, :]
        if cfg.mode == 0:
            f64, floor = f64 ** 2, floor ** 2
        with np.errstate(divide="ignore", invalid="ignore"):
            eg = np.abs(got["features"] - f64) / np.maximum(f64, floor)
            eo = np.abs(want["features"] - f64) / np.maximum(f64, floor)
        bad_vs_oracle = ~np.isclose(got["features"], want["features"], rtol=3e-5, atol=0)
        ok = bool((eg[bad_vs_oracle] < 1e-5).all() and (eg[bad_vs_oracle] <= eo[bad_vs_oracle]).all())
        n_f64 = globals().get("n_f64", 0) + 1
        if not ok:
            why.append(f"features: GPU vs float64 {eg[bad_vs_oracle].max():.3g}, oracle vs float64 {eo[bad_vs_oracle].max():.3g}")
    if aligned and cfg.decide == 1:   # occupancy away from the threshold must agree
        ref_f = want["features"][:, cfg.ref_band:cfg.ref_band + 1] if cfg.ref_band >= 0 else 1.0
        thr = np.array(cfg.thresh[:cfg.n_bands], np.float32)[None, :] * ref_f
        safe = np.abs(want["features"] / thr - 1) > 1e-4
        if not np.array_equal(got["occupancy"][safe], want["occupancy"][safe]):
            ok = False
            why.append("occupancy")
    if want_spec:
        truth = p64()
        fl = (1e-2 if cfg.frames_per_epoch >= 4 else 1e-1) * truth.mean(axis=1, keepdims=True)
        eg = (np.abs(got["spectrum"] - truth) / np.maximum(truth, fl)).max()
        eo = (np.abs(want["spectrum"] - truth) / np.maximum(truth, fl)).max()
        if not eg < 2 * eo + 2e-6:      # never further from float64 than the CPU restatement is (plus rounding headroom)
            ok = False
            why.append(f"spectrum: GPU vs float64 {eg:.3g}, oracle vs float64 {eo:.3g}")
    # wire format (int16 pairs) against the float path on the converted samples: bit for bit
    # (only a library with the optional wire-format kernels: $CRN_SENSE_LIB=.../libcrnsense_sc16.so; the measurement build has none)
    if ok and variant == 0 and rng.random() < 0.3 and cs.has_sc16():
        raw = np.clip(np.round(iq * 32768.0 * 40.0), -32768, 32767).astype(np.int16)      # x 40: a few hundred levels of noise
        fl = raw.astype(np.float32) / np.float32(32768.0)
        d_raw, d_fl = torch.from_numpy(raw).cuda(), torch.from_numpy(fl).cuda()
        outs2 = []
        for sc in (False, True):
            f_ = torch.zeros(n_epochs, cfg.n_bands, device="cuda")
            o_ = torch.zeros(n_epochs, cfg.n_bands, dtype=torch.uint8, device="cuda")
            d_ = torch.zeros(n_epochs, dtype=torch.int32, device="cuda")
            a_ = torch.zeros(n_epochs, 3, dtype=torch.float64, device="cuda")
            sp_ = torch.zeros(n_epochs, n, device="cuda") if want_spec else None
            s2 = cs.Sensor(cfg)
            s2.set_variant(deal)
            if epw:
                s2.set_variant(100 + epw)
            if tail >= 0:
                s2.set_variant(200 + tail)
            s2.run_device((d_raw if sc else d_fl).data_ptr(), n_epochs, L,
                          {"features": f_.data_ptr(), "ann_out": a_.data_ptr(), "decision": d_.data_ptr(), "occupancy": o_.data_ptr(),
                           "spectrum": sp_.data_ptr() if want_spec else 0}, epoch_stride=stride, sc16=sc)
            torch.cuda.synchronize()
            s2.close()
            outs2.append((f_, o_, d_, a_, sp_))
        ok = all(torch.equal(x, y) for x, y in zip(outs2[0], outs2[1]) if x is not None)
        if not ok:
            why.append("wire format differs from the float path")
        n_wire = globals().get("n_wire", 0) + 1
    if not ok:
        bad += 1
        print("MISMATCH seed", seed, dict(n=n, mode=cfg.mode, K=cfg.frames_per_epoch, win=cfg.window, L=L, variant=variant,
                                         spec=want_spec, epochs=n_epochs, ref_plan=ref_plan, welch=welch, aligned=aligned, nb=cfg.n_bands, decide=cfg.decide, epw=epw, tail=tail, stride=stride, deal=deal), why)
print(f"soak: {n_seeds} configurations, seeds {first} .. {first + n_seeds - 1} ({globals().get('n_wire', 0)} of them also through the wire-format path, compared bit for bit), {bad} mismatches"
      + (f"; {globals().get('n_f64', 0)} feature differences against the fp32 oracle were settled against float64 (GPU within 1e-5 of the truth at the per-bin floor, and closer than the oracle)" if globals().get('n_f64', 0) else ""))
sys.exit(1 if bad else 0)
