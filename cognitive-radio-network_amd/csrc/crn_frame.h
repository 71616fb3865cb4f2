// crn_frame.h — one frame of the sensing kernel: geometry, the IQ loads, the compile-time configuration, and the phases (three register
// passes, two LDS exchanges, the per-bin accumulate) that frame_compute strings together.
#ifndef CRN_FRAME_H
#define CRN_FRAME_H
#include "crn_butterflies.h"
#include "crn_kernels.h"

namespace crn {
// ---------------------------------------------------------------------------------------------
// geometry
// ---------------------------------------------------------------------------------------------
template <int R3>
struct Geo {
  static constexpr int N = 256 * R3;
  static constexpr int T = 16 * R3;           // threads per frame
  static constexpr int GROUPS = 256 / T;      // frame groups (epochs in flight) per workgroup
  static constexpr int ROW = T + R3;          // padded row of exchange 1 ([a][t]), complex units
  static constexpr int GROUP_CPLX = 16 * ROW; // one exchange buffer of one group
  static constexpr int J = 16 / R3;           // pass-3 butterflies per thread
  static constexpr bool XWAVE = (T > 64);     // exchange 1 crosses waves -> s_barrier needed
  static constexpr int TEAM = T < 64 ? T : 64;
};

__host__ __device__ constexpr int spec_phys(int k) { return k + (k >> 4); }  // padded float index

// IQ loads go through a buffer resource: the 128-bit descriptor and the per-frame / per-row part of
// the address live in SGPRs, each lane contributes one 32-bit byte offset, and reads past the end
// of the workgroup's window (ragged last workgroup, the prefetch after the last frame) return zero
// without touching memory.
typedef unsigned int v2u __attribute__((ext_vector_type(2)));

template <bool NT, bool SC = false>
CRN_DEV cx ld_iq(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff) {
  if constexpr (SC) {  // wire format: one dword = (int16 re, int16 im); kept raw until pass 1 consumes it (unpack_frame)
    const unsigned w = __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)voff, (int)soff, NT ? 2 : 0);
    return cx{__uint_as_float(w), 0.f};
  } else {
    const v2u v = __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)voff, (int)soff, NT ? 2 : 0);
    return cx{__uint_as_float(v.x), __uint_as_float(v.y)};
  }
}

// Wire-format samples (kSc16) become floats where a frame's registers are consumed: exactly what UHD's converter hands the
// reference's engine — int16 / 32768, both steps exact in fp32 — so every later bit is the bit the float path computes.
CRN_DEV void unpack_frame(cx (&u)[16]) {
#pragma unroll
  for (int r = 0; r < 16; r++) {
    const int w = (int)__float_as_uint(u[r].x);
    u[r] = cx{(float)(short)(w & 0xffff), (float)(w >> 16)};
  }
}
// The 1 / 32768 of that conversion is a power of two: it commutes with every rounding on the way (butterflies, |X|, the K-frame
// mean), so it is applied once per epoch where the accumulated sums leave the frame loop — 2^-15 on a sum of magnitudes,
// 2^-30 on a sum of energies — instead of twice per sample, and the results stay bit-identical to the float path's.
// (The constant comes with the launch — crn_sense_set_wire_full_scale — because converters differ: 2^-15 keeps the bit-identity,
// any other full scale gives the float path's results on floats converted with THAT constant to within rounding.)
template <class C>
CRN_DEV float sc_unscale(float x, const SenseParams &p) {
  if constexpr (C::SC16) return x * p.wire_unscale;
  else return x;
}

// u[r] = x[t + T r] of the frame that starts `frame_soff` bytes into the workgroup's window.
// Branch-free on purpose: a branch between issue and use makes the compiler drain vmcnt at the
// join, which serialises the prefetch with the compute it is meant to hide.
constexpr unsigned kOffNowhere = 0x80000000u;  // scalar offset past every window: the buffer range check drops the load

// Rows that lie wholly beyond the L samples a frame brings (short packets: the reference's 364 of
// 512, CE_Predictive_Node.cpp:149) are not fetched at all: they would be the next frame's samples.
template <int R3, bool NT, bool SC = false>
CRN_DEV void load_frame(cx (&u)[16], __amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned frame_soff, int L = Geo<R3>::N) {
  constexpr int T = Geo<R3>::T;
  constexpr int SB = SC ? 4 : 8;
#pragma unroll
  for (int r = 0; r < 16; r++) u[r] = ld_iq<NT, SC>(rsrc, voff, T * r < L ? frame_soff + (unsigned)(T * r * SB) : kOffNowhere);
}

// Half a frame: h[r] = x[t + T r], r = 0..7, of the N/2 samples starting `half_soff` bytes into the
// window (Welch mode: consecutive frames share a half, so each half is fetched once).
template <int R3, bool NT, bool SC = false>
CRN_DEV void load_half(cx (&h)[8], __amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned half_soff) {
  constexpr int T = Geo<R3>::T;
  constexpr int SB = SC ? 4 : 8;
#pragma unroll
  for (int r = 0; r < 8; r++) h[r] = ld_iq<NT, SC>(rsrc, voff, half_soff + (unsigned)(T * r * SB));
}

// Zero padding of a short frame (L < N), applied when the registers are consumed (reference: the
// FFT input buffer is zeroed once and only its first L entries are rewritten,
// CE_Predictive_Node.cpp:37,149).
template <int R3>
CRN_DEV void mask_frame(cx (&u)[16], int t, int L) {
  constexpr int T = Geo<R3>::T;
#pragma unroll
  for (int r = 0; r < 16; r++)
    if (t + T * r >= L) u[r] = cx{0.f, 0.f};
}

CRN_DEV void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// ---------------------------------------------------------------------------------------------
// Kernel configuration (compile time).
//   R3       N = 256 * R3
//   NBUF     LDS exchange buffers (2 drops the second barrier per frame when T > 64)
//   PREFETCH issue frame f+1's HBM loads before computing frame f (two register sets, ping-pong)
//   NT       nontemporal loads for the IQ stream
//   MAG      true: CRN_MODE_REF_MAG (|X|/K accumulate, feature = M^2); false: CRN_MODE_ENERGY
//   WIN      multiply by the window table
//   TW2LDS   pass-2 twiddles read from an LDS table instead of 30 registers
//   OCC      workgroups per CU the register allocation must allow
//   FULL     every frame brings all N samples (L == N): no zero-padding mask
//   PK       packed-f32 butterflies (see M<PK>)
// ---------------------------------------------------------------------------------------------
// OPT flags (the measurement build adds kTrace: crn_frame_ab.h)
enum : int {
  kSpread = 4,   // next frame's loads issued from inside passes 1 and 2, one per radix-4 group
  kLdsBlk = 32,  // LDS reads as hand-written ds_read_b64 blocks (no ds_read2_b64 merging)
  kTw1C = 64,    // pass-1 twiddles stored compressed (9 instead of 15 complex values)
  kRows = 256,   // pass 3 and the accumulate limited to the registers that can hold a bin of the reference channel plan (ref_acc_mask)
  kMulti = 512,  // a workgroup streams through several consecutive epoch groups
  kPrioValu = 1024, // s_setprio 1 through the butterflies of passes 1 and 2 (where the prefetch loads issue)
  kRegBands = 8192, // epoch close forms the band sums from registers (plans with n_row_entries > 0, no spectrum)
  kHannSym = 16384, // periodic Hann folded into pass 1's first butterflies (w[n + N/2] = 1 - w[n]): 8 window registers
  kTw2Early = 32768, // TW2LDS: the first block of pass-2 twiddles is read from LDS before the butterflies that precede its use
  kAlignedBands = 65536, // N = 4096, equal contiguous bands of 64 / 128 / 256 bins (p.aligned_shift): band sums by DPP + one barrier
  kSc16 = 131072,   // samples in HBM are the radio's wire format (two int16 per complex sample, 4 bytes): converted in pass 1
                    // (instantiated by crn_kernels_sc16.hip: a library built with make SC16=1)
  kDeal = 1048576,  // sense_kernel_dealt (launches of a few epochs): one epoch per workgroup, its frames dealt to the lane groups; pass 3
                    // parks each frame's per-bin values in LDS (ph_pass3_park) and the accumulate is replayed in frame order afterwards
};

template <int R3_, int NBUF_, bool PREFETCH_, bool NT_, bool MAG_, bool WIN_, bool TW2LDS_, int OCC_, bool FULL_, bool PK_, int OPT_ = 0>
struct Cfg {
  static constexpr int OPT = OPT_;  // OR of the flags above
  static constexpr int R3 = R3_, NBUF = NBUF_, OCC = OCC_;
  static constexpr bool PREFETCH = PREFETCH_, NT = NT_, MAG = MAG_, WIN = WIN_, TW2LDS = TW2LDS_, FULL = FULL_,
                        PK = PK_;
  static constexpr bool SC16 = (OPT_ & kSc16) != 0;
  static constexpr unsigned SB = SC16 ? 4u : 8u;       // bytes per complex sample in HBM
};

// Per-thread state that lives across the frames of an epoch.
template <class C>
struct FrameCtx {
  cx tw1[16];   // W_N^{t i}
  cx tw2[16];   // W_T^{m_lo i} (registers unless TW2LDS)
  float win[16];
  cx winp[4];   // kHannSym: (w[2p], w[2p + 1]) of rows 0..7
  float acc[16];
  const cx *tw2_lds;
  int wave;           // wave index in the workgroup (SGPR)
  int grp_epoch_stride;  // epoch of lane group g = epoch_base + g * this (1; the Welch stream deals epochs in runs)
  unsigned lds_base;  // LDS byte offset of the dynamic segment (SGPR); the band table copy sits behind tw2
  cx *gbuf;     // this group's exchange buffers
  int t, a, m_lo, L;
  unsigned park_off;  // kDeal: LDS byte offset of the slot the current frame's per-bin values go to
  float Kf, invK;
};

// Drops the next frame's loads into the current frame's butterfly stream one at a time: a wave
// that issues its 16 loads back to back sits on a full TA address FIFO for ~1000 cycles when HBM
// is near saturation (SQ_VMEM_TA_ADDR_FIFO_FULL), and being in-order it cannot compute meanwhile.
template <int R3, bool NT, bool SC = false>
struct SpreadLoads {
  cx (&nx)[16];
  __amdgpu_buffer_rsrc_t rsrc;
  unsigned voff, soff;
  int pass;   // 0 or 1: which of the frame's first two DFT16s this hook sits in
  bool half;  // Welch: only 8 loads (one half-frame), all in pass 1
  int L;      // samples a frame brings: rows wholly beyond it are not fetched
  __device__ __forceinline__ void operator()(int k) const {
    if (pass > 1 || (half && pass != 0)) return;
    const int idx = pass * 8 + k;
    __builtin_amdgcn_sched_barrier(0);
    nx[idx] = ld_iq<NT, SC>(rsrc, voff, Geo<R3>::T * idx < L ? soff + (unsigned)(Geo<R3>::T * idx * (SC ? 4 : 8)) : kOffNowhere);
    __builtin_amdgcn_sched_barrier(0);
  }
};



// ---- phases of one frame; `u` holds x[t + T r] on entry -------------------------------------
// pass 1: (zero pad, window,) DFT16 over r, twiddle W_N^{t a}
template <class C, class Hook = NoHook>
CRN_DEV void ph_pass1(cx (&u)[16], cx (&v)[16], FrameCtx<C> &c, const Hook &hook = Hook()) {
  using m = M<C::PK>;
  if constexpr (C::SC16) unpack_frame(u);
  if constexpr (!C::FULL) mask_frame<C::R3>(u, c.t, c.L);
  if constexpr (C::WIN && (C::OPT & kHannSym) != 0) {
    dft16_hann<C::PK>(u, v, c.winp, hook);
  } else {
    if constexpr (C::WIN) {
#pragma unroll
      for (int r = 0; r < 16; r++) u[r] = cx{u[r].x * c.win[r], u[r].y * c.win[r]};
    }
    dft16<C::PK>(u, v, hook);
  }
  if constexpr ((C::OPT & kTw1C) != 0) {
    // compressed table: tw1[1..8] = W^{t i}, tw1[0] = W^{16 t}; W^{t (16-i)} = W^{16 t} conj(W^{t i})
#pragma unroll
    for (int i = 1; i <= 8; i++) v[i] = m::mul(v[i], c.tw1[i]);
#pragma unroll
    for (int i = 9; i < 16; i++) v[i] = m::mul_conj(m::mul(v[i], c.tw1[0]), c.tw1[16 - i]);
  } else {
#pragma unroll
    for (int i = 1; i < 16; i++) v[i] = m::mul(v[i], c.tw1[i]);
  }
}
// exchange 1, layout [a][t] with rows of T + R3 complex
template <class C>
CRN_DEV void ph_x1_write(const cx (&v)[16], cx *buf, FrameCtx<C> &c) {
#pragma unroll
  for (int i = 0; i < 16; i++) buf[i * Geo<C::R3>::ROW + c.t] = v[i];
}
// Sixteen ds_read_b64 from one base address + immediate offsets, and the wait for them, as one
// asm block.  hipcc merges adjacent reads into ds_read2_b64, which moves half the bytes per LDS
// cycle of ds_read_b64 on gfx950 (MI355X_MICROARCH.md §LDS).
#define CRN_RD(i) "ds_read_b64 %" #i ", %16 offset:%" 
template <int STRIDE_BYTES>
CRN_DEV void lds_read16_b64(cx (&u)[16], const cx *base) {
  const unsigned addr = (unsigned)(size_t)base;  // LDS aperture: low 32 bits are the LDS byte address
  asm volatile(
      "ds_read_b64 %0, %16 offset:%17\n\tds_read_b64 %1, %16 offset:%18\n\tds_read_b64 %2, %16 offset:%19\n\t"
      "ds_read_b64 %3, %16 offset:%20\n\tds_read_b64 %4, %16 offset:%21\n\tds_read_b64 %5, %16 offset:%22\n\t"
      "ds_read_b64 %6, %16 offset:%23\n\tds_read_b64 %7, %16 offset:%24\n\tds_read_b64 %8, %16 offset:%25\n\t"
      "ds_read_b64 %9, %16 offset:%26\n\tds_read_b64 %10, %16 offset:%27\n\tds_read_b64 %11, %16 offset:%28\n\t"
      "ds_read_b64 %12, %16 offset:%29\n\tds_read_b64 %13, %16 offset:%30\n\tds_read_b64 %14, %16 offset:%31\n\t"
      "ds_read_b64 %15, %16 offset:%32\n\ts_waitcnt lgkmcnt(0)"
      : "=&v"(u[0]), "=&v"(u[1]), "=&v"(u[2]), "=&v"(u[3]), "=&v"(u[4]), "=&v"(u[5]), "=&v"(u[6]), "=&v"(u[7]), "=&v"(u[8]),
        "=&v"(u[9]), "=&v"(u[10]), "=&v"(u[11]), "=&v"(u[12]), "=&v"(u[13]), "=&v"(u[14]), "=&v"(u[15])
      : "v"(addr), "n"(0 * STRIDE_BYTES), "n"(1 * STRIDE_BYTES), "n"(2 * STRIDE_BYTES), "n"(3 * STRIDE_BYTES),
        "n"(4 * STRIDE_BYTES), "n"(5 * STRIDE_BYTES), "n"(6 * STRIDE_BYTES), "n"(7 * STRIDE_BYTES),
        "n"(8 * STRIDE_BYTES), "n"(9 * STRIDE_BYTES), "n"(10 * STRIDE_BYTES), "n"(11 * STRIDE_BYTES),
        "n"(12 * STRIDE_BYTES), "n"(13 * STRIDE_BYTES), "n"(14 * STRIDE_BYTES), "n"(15 * STRIDE_BYTES)
      : "memory");
}
#undef CRN_RD

// (Outputs are early-clobber: the address register must survive until the last read has issued.)
// Eight ds_read_b64 + wait as one block (pass-2 twiddles from the LDS table, two blocks per frame
// instead of the eight dependent read-wait-multiply round trips the compiler schedules).
template <int STRIDE_BYTES>
CRN_DEV void lds_read8_b64(cx (&w)[8], const cx *base) {
  const unsigned addr = (unsigned)(size_t)base;
  asm volatile(
      "ds_read_b64 %0, %8 offset:%9\n\tds_read_b64 %1, %8 offset:%10\n\tds_read_b64 %2, %8 offset:%11\n\t"
      "ds_read_b64 %3, %8 offset:%12\n\tds_read_b64 %4, %8 offset:%13\n\tds_read_b64 %5, %8 offset:%14\n\t"
      "ds_read_b64 %6, %8 offset:%15\n\tds_read_b64 %7, %8 offset:%16\n\ts_waitcnt lgkmcnt(0)"
      : "=&v"(w[0]), "=&v"(w[1]), "=&v"(w[2]), "=&v"(w[3]), "=&v"(w[4]), "=&v"(w[5]), "=&v"(w[6]), "=&v"(w[7])
      : "v"(addr), "n"(0 * STRIDE_BYTES), "n"(1 * STRIDE_BYTES), "n"(2 * STRIDE_BYTES), "n"(3 * STRIDE_BYTES),
        "n"(4 * STRIDE_BYTES), "n"(5 * STRIDE_BYTES), "n"(6 * STRIDE_BYTES), "n"(7 * STRIDE_BYTES)
      : "memory");
}

// The same eight reads without the wait (the caller consumes them after lds_wait8) ...
template <int STRIDE_BYTES>
CRN_DEV void lds_issue8_b64(cx (&w)[8], const cx *base) {
  const unsigned addr = (unsigned)(size_t)base;
  asm volatile(
      "ds_read_b64 %0, %8 offset:%9\n\tds_read_b64 %1, %8 offset:%10\n\tds_read_b64 %2, %8 offset:%11\n\t"
      "ds_read_b64 %3, %8 offset:%12\n\tds_read_b64 %4, %8 offset:%13\n\tds_read_b64 %5, %8 offset:%14\n\t"
      "ds_read_b64 %6, %8 offset:%15\n\tds_read_b64 %7, %8 offset:%16"
      : "=&v"(w[0]), "=&v"(w[1]), "=&v"(w[2]), "=&v"(w[3]), "=&v"(w[4]), "=&v"(w[5]), "=&v"(w[6]), "=&v"(w[7])
      : "v"(addr), "n"(0 * STRIDE_BYTES), "n"(1 * STRIDE_BYTES), "n"(2 * STRIDE_BYTES), "n"(3 * STRIDE_BYTES),
        "n"(4 * STRIDE_BYTES), "n"(5 * STRIDE_BYTES), "n"(6 * STRIDE_BYTES), "n"(7 * STRIDE_BYTES)
      : "memory");
}
// ... and the wait: the registers are tied to it so that no use is scheduled above it.
CRN_DEV void lds_wait8(cx (&w)[8]) {
  asm volatile("s_waitcnt lgkmcnt(0)"
               : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]), "+v"(w[4]), "+v"(w[5]), "+v"(w[6]), "+v"(w[7])
               :
               : "memory");
}

template <class C>
CRN_DEV void ph_x1_read(cx (&u)[16], cx *buf, FrameCtx<C> &c) {
  const cx *row = buf + c.a * Geo<C::R3>::ROW;
  if constexpr ((C::OPT & kLdsBlk) != 0) {
    lds_read16_b64<C::R3 * 8>(u, row + c.m_lo);
    return;
  }
#pragma unroll
  for (int i = 0; i < 16; i++) u[i] = row[C::R3 * i + c.m_lo];
}
// pass 2: DFT16 over m_hi, twiddle W_T^{m_lo c}
template <class C, class Hook = NoHook>
CRN_DEV void ph_pass2(cx (&u)[16], cx (&v)[16], FrameCtx<C> &c, const Hook &hook = Hook()) {
  using m = M<C::PK>;
  if constexpr (C::TW2LDS && (C::OPT & kLdsBlk) != 0 && (C::OPT & kTw2Early) != 0) {
    // rows 1..8 are in flight while the butterflies run; rows 8..15 while rows 1..8 are applied
    cx wa[8], wb[8];
    lds_issue8_b64<C::R3 * 8>(wa, c.tw2_lds + 1 * C::R3 + c.m_lo);
    dft16<C::PK>(u, v, hook);
    lds_wait8(wa);
    lds_issue8_b64<C::R3 * 8>(wb, c.tw2_lds + 8 * C::R3 + c.m_lo);
#pragma unroll
    for (int i = 1; i <= 8; i++) v[i] = m::mul(v[i], wa[i - 1]);
    lds_wait8(wb);
#pragma unroll
    for (int i = 9; i < 16; i++) v[i] = m::mul(v[i], wb[i - 8]);
    return;
  }
  dft16<C::PK>(u, v, hook);
  if constexpr (C::TW2LDS && (C::OPT & kLdsBlk) != 0) {
    cx w[8];
    lds_read8_b64<C::R3 * 8>(w, c.tw2_lds + 1 * C::R3 + c.m_lo);  // rows 1..8
#pragma unroll
    for (int i = 1; i <= 8; i++) v[i] = m::mul(v[i], w[i - 1]);
    lds_read8_b64<C::R3 * 8>(w, c.tw2_lds + 8 * C::R3 + c.m_lo);  // rows 8..15
#pragma unroll
    for (int i = 9; i < 16; i++) v[i] = m::mul(v[i], w[i - 8]);
    return;
  }
#pragma unroll
  for (int i = 1; i < 16; i++) v[i] = m::mul(v[i], C::TW2LDS ? c.tw2_lds[i * C::R3 + c.m_lo] : c.tw2[i]);
}
// exchange 2, inside the R3 lanes sharing `a`: slot (c, m) at c*R3 + m + c/J of the group's own row
template <class C>
CRN_DEV void ph_x2_write(const cx (&v)[16], cx *buf, FrameCtx<C> &c) {
  constexpr int R3 = C::R3, J = Geo<R3>::J;
  cx *row = buf + c.a * Geo<R3>::ROW;
#pragma unroll
  for (int cc = 0; cc < 16; cc++) row[cc * R3 + c.m_lo + cc / J] = v[cc];
}
template <class C>
CRN_DEV void ph_x2_read(cx (&u)[16], cx *buf, FrameCtx<C> &c) {
  constexpr int R3 = C::R3, J = Geo<R3>::J;
  const cx *row = buf + c.a * Geo<R3>::ROW;
  if constexpr ((C::OPT & kLdsBlk) != 0 && R3 == 16) {
    lds_read16_b64<8>(u, row + 17 * c.m_lo);
    return;
  }
  // thread (a, g = m_lo) takes c = g*J + j, all m
#pragma unroll
  for (int j = 0; j < J; j++)
#pragma unroll
    for (int mm = 0; mm < R3; mm++) u[j * R3 + mm] = row[(c.m_lo * J + j) * R3 + mm + c.m_lo];
}
// pass 3: DFT_R3 over m_lo -> d; bin k = a + 16 (g J + j) + 256 d; then the per-bin accumulate
// over the epoch (reference: fft_avg[i] += cabsf(X[i]) / K, CE_Predictive_Node.cpp:152-154)
template <class C>
CRN_DEV void ph_pass3(cx (&u)[16], cx (&v)[16]) {
  constexpr int R3 = C::R3, J = Geo<R3>::J;
  using m = M<C::PK>;
  if constexpr (R3 == 16) {
    dft16<C::PK>(u, v);
  } else if constexpr (R3 == 8) {
#pragma unroll
    for (int j = 0; j < J; j++) {
      cx in8[8], out8[8];
#pragma unroll
      for (int mm = 0; mm < 8; mm++) in8[mm] = u[j * 8 + mm];
      dft8<C::PK>(in8, out8);
#pragma unroll
      for (int mm = 0; mm < 8; mm++) v[j * 8 + mm] = out8[mm];
    }
  } else if constexpr (R3 == 4) {
#pragma unroll
    for (int j = 0; j < J; j++) {
      dft4<C::PK>(u[j * 4], u[j * 4 + 1], u[j * 4 + 2], u[j * 4 + 3]);
#pragma unroll
      for (int mm = 0; mm < 4; mm++) v[j * 4 + mm] = u[j * 4 + mm];
    }
  } else {
#pragma unroll
    for (int j = 0; j < J; j++) {
      v[j * 2] = m::add(u[j * 2], u[j * 2 + 1]);
      v[j * 2 + 1] = m::sub(u[j * 2], u[j * 2 + 1]);
    }
  }
}

// Accumulator registers a kernel keeps: all 16, or (kRows) those that can hold a bin of the reference channel plan.
template <class C>
constexpr unsigned acc_mask() {
  return (C::OPT & kRows) != 0 ? ref_acc_mask(C::R3) : 0xFFFFu;
}

// pass 3 restricted to the outputs named in MASK (bit j R3 + d); what it forms is what ph_pass3 forms, bit for bit
template <class C, unsigned MASK>
CRN_DEV void ph_pass3_pruned(cx (&u)[16], cx (&v)[16]) {
  constexpr int R3 = C::R3, J = Geo<R3>::J;
  using m = M<C::PK>;
  if constexpr (R3 == 16) {
    dft16_pruned<C::PK, MASK>(u, v);
  } else if constexpr (R3 == 8) {
    static_for<J>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      constexpr unsigned M8 = (MASK >> (j * 8)) & 0xFFu;
      if constexpr (M8 != 0) {
        cx in8[8], out8[8];
#pragma unroll
        for (int mm = 0; mm < 8; mm++) in8[mm] = u[j * 8 + mm];
        dft8_pruned<C::PK, M8>(in8, out8);
#pragma unroll
        for (int mm = 0; mm < 8; mm++)
          if ((M8 >> mm) & 1) v[j * 8 + mm] = out8[mm];
      }
    });
  } else if constexpr (R3 == 4) {
    static_for<J>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      constexpr unsigned M4 = (MASK >> (j * 4)) & 0xFu;
      if constexpr (M4 != 0) {
        dft4_pruned<C::PK, M4>(u[j * 4], u[j * 4 + 1], u[j * 4 + 2], u[j * 4 + 3]);
#pragma unroll
        for (int mm = 0; mm < 4; mm++) v[j * 4 + mm] = u[j * 4 + mm];
      }
    });
  } else {
    static_for<J>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      if constexpr (((MASK >> (j * 2)) & 1u) != 0) v[j * 2] = m::add(u[j * 2], u[j * 2 + 1]);
      if constexpr (((MASK >> (j * 2 + 1)) & 1u) != 0) v[j * 2 + 1] = m::sub(u[j * 2], u[j * 2 + 1]);
    });
  }
}

// one bin's contribution to its accumulator (reference: fft_avg[i] += cabsf(X[i]) / K, CE_Predictive_Node.cpp:152-154)
template <class C>
CRN_DEV void acc_bin(float &acc, cx x, float invK) {
  if constexpr (C::MAG) acc = fmaf(__builtin_amdgcn_sqrtf(fmaf(x.x, x.x, x.y * x.y)), invK, acc);
  else acc = fmaf(x.y, x.y, fmaf(x.x, x.x, acc));
}

// kDeal: pass 3, then the frame's per-bin values — |X| (MAG) or X itself (energy mode: the accumulate is a chain of two fmas on the
// parts) — go to the frame's LDS slot, [16 registers][T threads], instead of into the accumulators: the workgroup's lane groups
// work on different frames of ONE epoch, and the K-frame accumulate is replayed from the slots in frame order (replay_parked), so
// the sums are bit for bit the streaming kernel's.
typedef __attribute__((address_space(3))) float lds_park_f32;
typedef __attribute__((address_space(3))) cx lds_park_cx;
template <class C>
CRN_DEV void ph_pass3_park(cx (&u)[16], FrameCtx<C> &c) {
  constexpr int T = Geo<C::R3>::T;
  cx v[16];
  ph_pass3<C>(u, v);
  if constexpr (C::MAG) {
    lds_park_f32 *slot = reinterpret_cast<lds_park_f32 *>(c.park_off);
#pragma unroll
    for (int i = 0; i < 16; i++) slot[i * T + c.t] = __builtin_amdgcn_sqrtf(fmaf(v[i].x, v[i].x, v[i].y * v[i].y));
  } else {
    lds_park_cx *slot = reinterpret_cast<lds_park_cx *>(c.park_off);
#pragma unroll
    for (int i = 0; i < 16; i++) slot[i * T + c.t] = v[i];
  }
}
// The accumulate of ph_pass3_acc over frames [0, K) from their slots (slot f at park_base + f slot_bytes), same operations, same order.
template <class C>
CRN_DEV void replay_parked(FrameCtx<C> &c, unsigned park_base, int K) {
  constexpr int T = Geo<C::R3>::T;
  constexpr unsigned kSlotBytes = (unsigned)Geo<C::R3>::N * (C::MAG ? 4u : 8u);
  for (int f = 0; f < K; f++) {
    const unsigned off = park_base + (unsigned)f * kSlotBytes;
#pragma unroll
    for (int i = 0; i < 16; i++) {
      if constexpr (C::MAG) {
        const float mag = reinterpret_cast<const lds_park_f32 *>(off)[i * T + c.t];
        c.acc[i] = fmaf(mag, c.invK, c.acc[i]);
      } else {
        const cx x = reinterpret_cast<const lds_park_cx *>(off)[i * T + c.t];
        c.acc[i] = fmaf(x.y, x.y, fmaf(x.x, x.x, c.acc[i]));
      }
    }
  }
}

// pass 3 + per-bin accumulate: v[j * R3 + d] is bin a + 16 (m_lo J + j) + 256 d
template <class C>
CRN_DEV void ph_pass3_acc(cx (&u)[16], FrameCtx<C> &c) {
  if constexpr ((C::OPT & kDeal) != 0) {
    ph_pass3_park<C>(u, c);
    return;
  }
  cx v[16];
  constexpr unsigned MASK = acc_mask<C>();
  if constexpr (MASK == 0xFFFFu) {
    ph_pass3<C>(u, v);
  } else {
#pragma unroll
    for (int i = 0; i < 16; i++) v[i] = cx{0.f, 0.f};
    ph_pass3_pruned<C, MASK>(u, v);
  }
#pragma unroll
  for (int i = 0; i < 16; i++) {
    if (((MASK >> i) & 1u) == 0) continue;   // a register no band of the plan reaches: neither formed nor accumulated
    if constexpr (C::MAG) {
      // |X| / K per frame.  v_sqrt_f32 (1 ulp) and a multiply by 1/K instead of the reference's
      // correctly rounded hypotf and divide: each addend moves by <= 2 ulp, five orders of
      // magnitude inside the 1e-5 feature tolerance, at a fifth of the instructions.
      const float mag = __builtin_amdgcn_sqrtf(fmaf(v[i].x, v[i].x, v[i].y * v[i].y));
      c.acc[i] = fmaf(mag, c.invK, c.acc[i]);
    } else {
      c.acc[i] = fmaf(v[i].y, v[i].y, fmaf(v[i].x, v[i].x, c.acc[i]));
    }
  }
}

template <class C>
CRN_DEV void group_sync() {
  if constexpr (Geo<C::R3>::XWAVE) __syncthreads();
  else wave_sync();
}

// One frame: three register passes + two LDS exchanges + per-bin accumulate.  `u` is clobbered.  The next frame (`nx`, at
// `soff_next` in the workgroup's window; HALF: its new half-frame, the Welch stream) is fetched from inside passes 1 and 2.
template <class C, bool HALF = false>
CRN_DEV void frame_compute(cx (&u)[16], FrameCtx<C> &c, int f, cx (&nx)[16], __amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff_next) {
  using G = Geo<C::R3>;
  static_assert((C::OPT & kSpread) != 0, "the kernels fetch the next frame from inside the current one's butterflies");
  cx *buf = c.gbuf + (C::NBUF == 2 ? (f & 1) * G::GROUP_CPLX : 0);
  cx v[16];
  const int Lrows = C::FULL ? G::N : c.L;
  const SpreadLoads<C::R3, C::NT, C::SC16> h1{nx, rsrc, voff, soff_next, 0, HALF, Lrows}, h2{nx, rsrc, voff, soff_next, 1, HALF, Lrows};
  // Waves in passes 1 and 2 (which also issue the next frame's loads) win VALU arbitration
  // against waves in pass 3 / epoch close: measured +1.4 % (76.9 vs 75.8 %); raising pass 1 alone,
  // pass 3 alone or the LDS phases gains nothing.
  constexpr bool PV = (C::OPT & kPrioValu) != 0;
  if constexpr (PV) __builtin_amdgcn_s_setprio(1);
  ph_pass1<C>(u, v, c, h1);
  if constexpr (PV) __builtin_amdgcn_s_setprio(0);
  if constexpr (G::XWAVE && C::NBUF == 1) __syncthreads();  // rows may still be read as exchange 2
  ph_x1_write<C>(v, buf, c);
  group_sync<C>();
  ph_x1_read<C>(u, buf, c);
  if constexpr (PV) __builtin_amdgcn_s_setprio(1);
  ph_pass2<C>(u, v, c, h2);
  if constexpr (PV) __builtin_amdgcn_s_setprio(0);
  wave_sync();
  ph_x2_write<C>(v, buf, c);
  wave_sync();
  ph_x2_read<C>(u, buf, c);
  ph_pass3_acc<C>(u, c);
}

}  // namespace crn
#endif
