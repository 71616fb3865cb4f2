"""The transform on its own: crn_fft_forward_device, and liquid-dsp's fft_create_plan / fft_execute /
fft_destroy_plan over it (include/crn_liquid_fft.h) — the entry points the reference engine binds
(CE_Predictive_Node.cpp:42-45,150)."""
import ctypes as C
import os

import numpy as np
import pytest

import crnsense as cs
import oracle_py as orc


def test_shim_library_exports_the_liquid_entry_points(built):
    """Every function include/crn_liquid_fft.h declares is exported by libcrnliquidfft.so (loads without a GPU)."""
    import re
    assert os.path.exists(cs.LIQUID_SHIM_PATH)
    L = C.CDLL(cs.LIQUID_SHIM_PATH)
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "crn_liquid_fft.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)                      # prototypes only, not the prose
    declared = set(re.findall(r"\b(fft_[a-z_]+|crn_liquid_[a-z_]+)\s*\(", hdr))
    assert declared == {"fft_create_plan", "fft_execute", "fft_destroy_plan", "crn_liquid_fft_forwarded"}, declared
    for name in declared:
        assert hasattr(L, name), name


def _rel_err(x, ref):
    return np.abs(x - ref).max() / np.sqrt((np.abs(ref) ** 2).mean())


@pytest.mark.gpu
@pytest.mark.parametrize("n", [512, 1024, 2048, 4096])
def test_forward_fft_matches_float64(built, n):
    """Complex spectrum against numpy's complex128 FFT: whole frames, zero-padded short frames, a frame
    stride with gaps, a batch that leaves the last workgroup ragged.  Never further from float64 than
    the liquid-style CPU restatement is (plus headroom), and inside 1e-5 of the spectrum's RMS."""
    import torch
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(n)
    s = cs.Sensor(cs.cfg_energy_scaled(n, 4.0))
    for L, stride, frames in ((n, 0, 37), (364 if n == 512 else n - 3, 0, 9), (n // 2, n + 5, 11), (1, 0, 3)):
        st = stride or L
        x = (rng.standard_normal(frames * st + n) + 1j * rng.standard_normal(frames * st + n)).astype(np.complex64)
        x[5::97] += 3.0  # some structure besides noise
        d_in = torch.from_numpy(x.view(np.float32)).to(dev)
        d_out = torch.zeros(frames * n * 2, dtype=torch.float32, device=dev)
        s.fft_forward_device(d_in.data_ptr(), frames, L, d_out.data_ptr(), frame_stride=stride)
        torch.cuda.synchronize()
        got = d_out.cpu().numpy().view(np.complex64).reshape(frames, n)
        for f in range(frames):
            padded = np.zeros(n, np.complex128)
            padded[:L] = x[f * st:f * st + L]
            ref = np.fft.fft(padded)
            e_gpu = _rel_err(got[f], ref)
            e_cpu = _rel_err(orc.fft_radix2(padded.astype(np.complex64)), ref)
            assert e_gpu < 1e-5 and e_gpu < 2.0 * e_cpu + 1e-6, (L, stride, f, e_gpu, e_cpu)
    s.close()


@pytest.mark.gpu
def test_forward_fft_argument_errors(built):
    import torch
    dev = torch.device("cuda", 0)
    s = cs.Sensor(cs.cfg_reference())
    buf = torch.zeros(4096, dtype=torch.float32, device=dev)
    with pytest.raises(cs.CrnError):
        s.fft_forward_device(buf.data_ptr(), 1, 513, buf.data_ptr())
    with pytest.raises(cs.CrnError):
        s.fft_forward_device(buf.data_ptr(), -1, 512, buf.data_ptr())
    with pytest.raises(cs.CrnError):
        s.fft_forward_device(0, 1, 512, buf.data_ptr())
    s.fft_forward_device(buf.data_ptr(), 0, 512, buf.data_ptr())  # nothing to do is not an error
    s.close()


@pytest.mark.gpu
@pytest.mark.parametrize("n", [512, 4096])
def test_liquid_entry_points_the_way_the_reference_uses_them(built, n):
    """fft_create_plan binds the caller's arrays (CE_Predictive_Node.cpp:42-45: member arrays buffer /
    buffer_F); every fft_execute transforms what is in x *now* into y (.cpp:149-150: memcpy a packet
    into the zeroed buffer, execute)."""
    L = C.CDLL(cs.LIQUID_SHIM_PATH)
    L.fft_create_plan.restype = C.c_void_p
    L.fft_create_plan.argtypes = [C.c_uint, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    L.fft_execute.argtypes = [C.c_void_p]
    L.fft_destroy_plan.argtypes = [C.c_void_p]
    x = np.zeros(n, np.complex64)
    y = np.zeros(n, np.complex64)
    plan = L.fft_create_plan(n, x.ctypes.data, y.ctypes.data, 1, 0)
    assert plan
    rng = np.random.default_rng(7)
    for packet_len in (n, 364, 363):
        packet_len = min(packet_len, n)
        x[:] = 0
        x[:packet_len] = (rng.standard_normal(packet_len) + 1j * rng.standard_normal(packet_len)).astype(np.complex64)
        L.fft_execute(plan)
        ref = np.fft.fft(x.astype(np.complex128))
        assert _rel_err(y, ref) < 1e-5
        assert np.abs(y - orc.fft_radix2(x)).max() / np.sqrt((np.abs(ref) ** 2).mean()) < 1e-5
    L.fft_destroy_plan(plan)


@pytest.mark.gpu
def test_reference_epoch_loop_over_the_liquid_entry_points(built):
    """The reference's own loop (CE_Predictive_Node.cpp:148-261) written out around fft_execute — what
    the unchanged engine would do when it links libcrnliquidfft instead of liquid's FFT: per packet
    memcpy into the zeroed 512 buffer, fft_execute, fft_avg += |X|/10; per 10 packets band sums,
    squares, network, cascade.  Same decisions and features as the oracle's literal epoch."""
    import signals
    import time
    lib = C.CDLL(cs.LIQUID_SHIM_PATH)
    lib.fft_create_plan.restype = C.c_void_p
    lib.fft_create_plan.argtypes = [C.c_uint, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    lib.fft_execute.argtypes = [C.c_void_p]
    lib.fft_destroy_plan.argtypes = [C.c_void_p]
    cfg = cs.cfg_reference()
    L, n_epochs = 364, 8
    iq, picks = signals.make_epochs(cfg, n_epochs, seed=99, L=L)
    pk = iq.view(np.complex64).reshape(n_epochs, 10, L)
    buf = np.zeros(512, np.complex64)
    buf_f = np.zeros(512, np.complex64)
    plan = lib.fft_create_plan(512, buf.ctypes.data, buf_f.ctypes.data, 1, 0)
    t_exec = []
    for e in range(n_epochs):
        fft_avg = np.zeros(512, np.float32)
        for f in range(10):
            buf[:L] = pk[e, f]                                    # .cpp:149
            t0 = time.perf_counter()
            lib.fft_execute(plan)                                 # .cpp:150
            t_exec.append(time.perf_counter() - t0)
            fft_avg += np.abs(buf_f).astype(np.float32) / np.float32(10)   # .cpp:152-154
        m1 = np.float32(0)
        for k in list(range(0, 16)) + list(range(496, 511)):      # .cpp:173-179 (bin 511 excluded)
            m1 += fft_avg[k]
        m2 = np.float32(0)
        for k in range(55, 85):
            m2 += fft_avg[k]
        m3 = np.float32(0)
        for k in range(189, 222):
            m3 += fft_avg[k]
        nf = np.float32(0)
        for k in range(300, 310):
            nf += fft_avg[k]
        feat = np.array([nf * nf, m1 * m1, m2 * m2, m3 * m3], np.float32)   # .cpp:194-200
        d, out3 = orc.ann(feat)                                   # .cpp:214-261
        ref = orc.ref_epoch(iq[e * 10 * L * 2:(e + 1) * 10 * L * 2], L)
        assert d == ref["decision"] == picks[e]
        assert np.allclose(feat, ref["features"], rtol=1e-5)
        assert np.abs(out3 - ref["ann_out"]).max() < 1e-6
    lib.fft_destroy_plan(plan)
    print(f"fft_execute over the shim: median {np.median(t_exec) * 1e6:.0f} us per 512-point call")


@pytest.mark.gpu
@pytest.mark.parametrize("n", [512, 1024, 2048, 4096])
def test_forward_fft_against_rocfft(built, n):
    """Third, independent cross-check (test side only; the product never links rocFFT): torch.fft.fft on
    the same device buffer, complex64, which is rocFFT's plan for that size."""
    import torch
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(n)
    frames = 64
    x = torch.randn(frames, n, 2, generator=g, device=dev, dtype=torch.float32)
    out = torch.zeros(frames, n, 2, device=dev, dtype=torch.float32)
    s = cs.Sensor(cs.cfg_energy_scaled(n, 4.0))
    s.fft_forward_device(x.data_ptr(), frames, n, out.data_ptr())
    torch.cuda.synchronize()
    s.close()
    ref = torch.fft.fft(torch.view_as_complex(x), dim=1)
    got = torch.view_as_complex(out)
    rms = ref.abs().pow(2).mean().sqrt()
    assert ((got - ref).abs().max() / rms).item() < 1e-5
    ref64 = torch.fft.fft(torch.view_as_complex(x.double()), dim=1)
    ours = ((got.to(torch.complex128) - ref64).abs().max() / rms).item()
    theirs = ((ref.to(torch.complex128) - ref64).abs().max() / rms).item()
    assert ours < 2.0 * theirs + 1e-6, (ours, theirs)   # as close to float64 as the vendor library
