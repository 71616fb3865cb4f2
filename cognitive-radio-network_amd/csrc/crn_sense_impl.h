// crn_sense_impl.h — the sensing kernel template and its launch helpers (everything up to the variant tables): included by
// crn_kernels.hip (complex-float input: variants, dispatch, the other kernels) and crn_kernels_sc16.hip (wire-format input), so
// that the two sets of instantiations compile in parallel.  Internal linkage throughout (static / constexpr / templates).
#ifndef CRN_SENSE_IMPL_H
#define CRN_SENSE_IMPL_H
#include <hip/hip_runtime.h>
#include <type_traits>
#include <stdint.h>

#include "crn_kernels.h"

namespace crn {

// A complex fp32 value lives in an even-aligned VGPR pair (re, im) so the packed-f32 VALU forms
// (v_pk_add/mul/fma_f32) work on it directly.
typedef float cx __attribute__((ext_vector_type(2)));
#define CRN_DEV static __device__ __forceinline__

// ---------------------------------------------------------------------------------------------
// Complex arithmetic, forward transform convention W = exp(-j theta).
//
// PK = true: one VOP3P instruction per complex add / rotate-add and two per complex multiply,
// with the re/im swaps and sign flips expressed through op_sel / neg modifiers, so no v_mov is
// spent on shuffling.  A lone wave issues one VALU instruction every ~4.6 cycles on gfx950
// whether it is packed or not (measured, tools/valu_rate.hip), so at the 3-4 waves per SIMD this
// kernel runs at, halving the instruction count is what shortens a frame.
// PK = false: plain scalar fp32 (reference build of the same arithmetic, used for A/B).
// Operand semantics (VOP3P, 64-bit sources): op_sel[i] picks the half of source i feeding the
// LOW result, op_sel_hi[i] the half feeding the HIGH result; neg_lo / neg_hi negate source i for
// the low / high result.
// ---------------------------------------------------------------------------------------------
template <bool PK>
struct M {
  CRN_DEV cx add(cx a, cx b) {
    if constexpr (PK) { cx d; asm("v_pk_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
    else return cx{a.x + b.x, a.y + b.y};
  }
  CRN_DEV cx sub(cx a, cx b) {
    if constexpr (PK) { cx d; asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b)); return d; }
    else return cx{a.x - b.x, a.y - b.y};
  }
  // a + (-j) b = (a.x + b.y, a.y - b.x)
  CRN_DEV cx add_mj(cx a, cx b) {
    if constexpr (PK) { cx d; asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b)); return d; }
    else return cx{a.x + b.y, a.y - b.x};
  }
  // a - (-j) b = (a.x - b.y, a.y + b.x)
  CRN_DEV cx sub_mj(cx a, cx b) {
    if constexpr (PK) { cx d; asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(d) : "v"(a), "v"(b)); return d; }
    else return cx{a.x - b.y, a.y + b.x};
  }
  // a * w, w in VGPRs (per-lane twiddle)
  CRN_DEV cx mul(cx a, cx w) {
    if constexpr (PK) {
      cx t, d;
      asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(t) : "v"(a), "v"(w));  // (a.x w.x, a.y w.x)
      asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]"
          : "=v"(d) : "v"(a), "v"(w), "v"(t));                                     // (-a.y w.y + t.x, a.x w.y + t.y)
      return d;
    } else {
      return cx{fmaf(-a.y, w.y, a.x * w.x), fmaf(a.y, w.x, a.x * w.y)};
    }
  }
  // a * conj(w) = (a.x w.x + a.y w.y, a.y w.x - a.x w.y)
  CRN_DEV cx mul_conj(cx a, cx w) {
    if constexpr (PK) {
      cx t, d;
      asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(t) : "v"(a), "v"(w));
      asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]"
          : "=v"(d) : "v"(a), "v"(w), "v"(t));
      return d;
    } else {
      return cx{fmaf(a.y, w.y, a.x * w.x), fmaf(-a.x, w.y, a.y * w.x)};
    }
  }
  // w_S * t + x (NEG: w_S * t - x), w_S = half S of the register pair wp: a real weight applied to a complex value
  template <int S, bool NEG>
  CRN_DEV cx fma_w(cx wp, cx t, cx x) {
    if constexpr (PK) {
      cx d;
      if constexpr (S == 0 && !NEG) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "=v"(d) : "v"(wp), "v"(t), "v"(x));
      if constexpr (S == 1 && !NEG) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "=v"(d) : "v"(wp), "v"(t), "v"(x));
      if constexpr (S == 0 && NEG) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_lo:[0,0,1] neg_hi:[0,0,1]" : "=v"(d) : "v"(wp), "v"(t), "v"(x));
      if constexpr (S == 1 && NEG) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1] neg_lo:[0,0,1] neg_hi:[0,0,1]" : "=v"(d) : "v"(wp), "v"(t), "v"(x));
      return d;
    } else {
      const float w = S == 0 ? wp.x : wp.y;
      return NEG ? cx{fmaf(w, t.x, -x.x), fmaf(w, t.y, -x.y)} : cx{fmaf(w, t.x, x.x), fmaf(w, t.y, x.y)};
    }
  }
  // a * w, w a wave-uniform constant held in an SGPR pair
  CRN_DEV cx mul_c(cx a, cx w) {
    if constexpr (PK) {
      cx t, d;
      asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(t) : "v"(a), "s"(w));
      asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]"
          : "=v"(d) : "v"(a), "s"(w), "v"(t));
      return d;
    } else {
      return cx{fmaf(-a.y, w.y, a.x * w.x), fmaf(a.y, w.x, a.x * w.y)};
    }
  }
};

// compile-time loop: f(std::integral_constant<int, 0>{}) ... f(std::integral_constant<int, N - 1>{})
template <int I, int N, class F>
CRN_DEV void static_for_(F &f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for_<I + 1, N>(f);
  }
}
template <int N, class F>
CRN_DEV void static_for(F &&f) {
  static_for_<0, N>(f);
}

#define CRN_C1 0.92387953251128674f  // cos(pi/8)
#define CRN_S1 0.38268343236508977f  // sin(pi/8)
#define CRN_H 0.70710678118654752f   // sqrt(1/2)

// 4-point forward DFT in place.  B2MJ: input a2 still lacks a factor -j (a folded W16^4 / W8^2).
template <bool PK, bool B2MJ = false>
CRN_DEV void dft4(cx &a0, cx &a1, cx &a2, cx &a3) {
  using m = M<PK>;
  const cx s02 = B2MJ ? m::add_mj(a0, a2) : m::add(a0, a2);
  const cx d02 = B2MJ ? m::sub_mj(a0, a2) : m::sub(a0, a2);
  const cx s13 = m::add(a1, a3), d13 = m::sub(a1, a3);
  a0 = m::add(s02, s13);
  a2 = m::sub(s02, s13);
  a1 = m::add_mj(d02, d13);  // d02 - j d13
  a3 = m::sub_mj(d02, d13);  // d02 + j d13
}

struct NoHook {
  __device__ __forceinline__ void operator()(int) const {}
};

// Twiddles W16^{r0 a0} and level B of the 16-point transform (shared by the plain and the windowed level A).
template <bool PK, class Hook = NoHook>
CRN_DEV void dft16_level_b(cx (&y)[16], cx (&out)[16], const Hook &hook = Hook()) {
  using m = M<PK>;
  // W16^{r0 a0} on element (r0, a0) = y[r0 + 4 a0]; W16^4 = -j is folded into level B
  const cx w1 = {CRN_C1, -CRN_S1}, w2 = {CRN_H, -CRN_H}, w3 = {CRN_S1, -CRN_C1};
  const cx w6 = {-CRN_H, -CRN_H}, w9 = {-CRN_C1, CRN_S1};
  y[1 + 4 * 1] = m::mul_c(y[1 + 4 * 1], w1);
  y[1 + 4 * 2] = m::mul_c(y[1 + 4 * 2], w2);
  y[1 + 4 * 3] = m::mul_c(y[1 + 4 * 3], w3);
  y[2 + 4 * 1] = m::mul_c(y[2 + 4 * 1], w2);
  y[2 + 4 * 3] = m::mul_c(y[2 + 4 * 3], w6);
  y[3 + 4 * 1] = m::mul_c(y[3 + 4 * 1], w3);
  y[3 + 4 * 2] = m::mul_c(y[3 + 4 * 2], w6);
  y[3 + 4 * 3] = m::mul_c(y[3 + 4 * 3], w9);
  // level B: for each a0, DFT4 over r0; X[a0 + 4 a1] = y[a1 + 4 a0]
  dft4<PK>(y[0], y[1], y[2], y[3]);
  hook(4);
  dft4<PK>(y[4], y[5], y[6], y[7]);
  hook(5);
  dft4<PK, true>(y[8], y[9], y[10], y[11]);  // y[10] carries the folded -j
  hook(6);
  dft4<PK>(y[12], y[13], y[14], y[15]);
  hook(7);
#pragma unroll
  for (int a0 = 0; a0 < 4; a0++)
#pragma unroll
    for (int a1 = 0; a1 < 4; a1++) out[a0 + 4 * a1] = y[a1 + 4 * a0];
}

// 16-point forward DFT, natural order in and out, as 4 x 4.  `hook(k)`, k = 0..7, runs after the
// k-th radix-4 group: the caller uses it to drop one prefetch load into the butterfly stream.
template <bool PK, class Hook = NoHook>
CRN_DEV void dft16(const cx (&in)[16], cx (&out)[16], const Hook &hook = Hook()) {
  cx y[16];
#pragma unroll
  for (int i = 0; i < 16; i++) y[i] = in[i];
  // level A: for each r0, DFT4 over r1 (r = r0 + 4 r1); a0 replaces r1
#pragma unroll
  for (int r0 = 0; r0 < 4; r0++) {
    dft4<PK>(y[r0], y[r0 + 4], y[r0 + 8], y[r0 + 12]);
    hook(r0);
  }
  dft16_level_b<PK>(y, out, hook);
}

// The same transform with a periodic Hann window folded into level A.  Rows r and r + 8 of a thread are
// samples n and n + N/2, where the window satisfies w[n + N/2] = 1 - w[n], so the first butterfly of the
// pair (x_lo, x_hi) needs one weight:
//   w x_lo + (1 - w) x_hi = w (x_lo - x_hi) + x_hi        w x_lo - (1 - w) x_hi = w (x_lo + x_hi) - x_hi
// — two packed adds and two packed FMAs per pair, the cycles of the 4 multiplies + 2 adds they replace in
// a third fewer instructions, and 8 window registers instead of 16.  wp[p] = (w[2p], w[2p + 1]), rows 0..7.
template <bool PK, class Hook = NoHook>
CRN_DEV void dft16_hann(const cx (&in)[16], cx (&out)[16], const cx (&wp)[4], const Hook &hook = Hook()) {
  using m = M<PK>;
  cx y[16];
#pragma unroll
  for (int i = 0; i < 16; i++) y[i] = in[i];
  static_for<4>([&](auto rc) {
    constexpr int r0 = decltype(rc)::value;
    constexpr int S = r0 & 1;
    const cx x0 = y[r0], x1 = y[r0 + 4], x2 = y[r0 + 8], x3 = y[r0 + 12];
    const cx w0 = wp[r0 / 2], w1 = wp[(r0 + 4) / 2];
    const cx s02 = m::template fma_w<S, false>(w0, m::sub(x0, x2), x2);
    const cx d02 = m::template fma_w<S, true>(w0, m::add(x0, x2), x2);
    const cx s13 = m::template fma_w<S, false>(w1, m::sub(x1, x3), x3);
    const cx d13 = m::template fma_w<S, true>(w1, m::add(x1, x3), x3);
    y[r0] = m::add(s02, s13);
    y[r0 + 8] = m::sub(s02, s13);
    y[r0 + 4] = m::add_mj(d02, d13);
    y[r0 + 12] = m::sub_mj(d02, d13);
    hook(r0);
  });
  dft16_level_b<PK>(y, out, hook);
}

// The reference hard-codes its channel plan (bins 0-15 + 496-510, 55-84, 189-221, 300-309 of 512:
// CE_Predictive_Node.cpp:173-191).  At N = 4096 those bands touch 7 of the 16 blocks of 256 bins, and
// the last radix-4 level of pass 3 produces exactly one block per output: row d = bins
// [256 d, 256 d + 256).  For band tables inside these rows, and when no per-bin spectrum is asked
// for, pass 3 forms and accumulates only the needed outputs (bit-identical for those bins).
static constexpr unsigned kRefPlanRows = 0x8267u;  // rows {0, 1, 2, 5, 6, 9, 15}

// DFT16 whose last level only forms the outputs named in MASK (bit d = X[d] needed).
template <bool PK, unsigned MASK>
CRN_DEV void dft16_pruned(const cx (&in)[16], cx (&out)[16]) {
  using m = M<PK>;
  cx y[16];
#pragma unroll
  for (int i = 0; i < 16; i++) y[i] = in[i];
#pragma unroll
  for (int r0 = 0; r0 < 4; r0++) dft4<PK>(y[r0], y[r0 + 4], y[r0 + 8], y[r0 + 12]);
  const cx w1 = {CRN_C1, -CRN_S1}, w2 = {CRN_H, -CRN_H}, w3 = {CRN_S1, -CRN_C1};
  const cx w6 = {-CRN_H, -CRN_H}, w9 = {-CRN_C1, CRN_S1};
  y[1 + 4 * 1] = m::mul_c(y[1 + 4 * 1], w1);
  y[1 + 4 * 2] = m::mul_c(y[1 + 4 * 2], w2);
  y[1 + 4 * 3] = m::mul_c(y[1 + 4 * 3], w3);
  y[2 + 4 * 1] = m::mul_c(y[2 + 4 * 1], w2);
  y[2 + 4 * 3] = m::mul_c(y[2 + 4 * 3], w6);
  y[3 + 4 * 1] = m::mul_c(y[3 + 4 * 1], w3);
  y[3 + 4 * 2] = m::mul_c(y[3 + 4 * 2], w6);
  y[3 + 4 * 3] = m::mul_c(y[3 + 4 * 3], w9);
#pragma unroll
  for (int a0 = 0; a0 < 4; a0++) {
    constexpr unsigned M0 = MASK;
    const bool n0 = (M0 >> (a0 + 0)) & 1, n1 = (M0 >> (a0 + 4)) & 1, n2 = (M0 >> (a0 + 8)) & 1, n3 = (M0 >> (a0 + 12)) & 1;
    const cx b0 = y[4 * a0], b1 = y[4 * a0 + 1], b2 = y[4 * a0 + 2], b3 = y[4 * a0 + 3];
    cx s02 = b0, d02 = b0, s13 = b1, d13 = b1;
    if (n0 || n2) { s02 = a0 == 2 ? m::add_mj(b0, b2) : m::add(b0, b2); s13 = m::add(b1, b3); }
    if (n1 || n3) { d02 = a0 == 2 ? m::sub_mj(b0, b2) : m::sub(b0, b2); d13 = m::sub(b1, b3); }
    if (n0) out[a0 + 0] = m::add(s02, s13);
    if (n1) out[a0 + 4] = m::add_mj(d02, d13);
    if (n2) out[a0 + 8] = m::sub(s02, s13);
    if (n3) out[a0 + 12] = m::sub_mj(d02, d13);
  }
}

// 8-point forward DFT as 2 x 4.
template <bool PK>
CRN_DEV void dft8(const cx (&in)[8], cx (&out)[8]) {
  using m = M<PK>;
  cx y[8];
#pragma unroll
  for (int i = 0; i < 8; i++) y[i] = in[i];
  dft4<PK>(y[0], y[2], y[4], y[6]);  // r = r0 + 2 r1: DFT4 over r1 -> a0 at y[r0 + 2 a0]
  dft4<PK>(y[1], y[3], y[5], y[7]);
  const cx w1 = {CRN_H, -CRN_H}, w3 = {-CRN_H, -CRN_H};
  y[3] = m::mul_c(y[3], w1);  // W8^1 on (r0 = 1, a0 = 1)
  y[7] = m::mul_c(y[7], w3);  // W8^3 on (r0 = 1, a0 = 3); W8^2 = -j on y[5] folded below
#pragma unroll
  for (int a0 = 0; a0 < 4; a0++) {
    const cx e = y[2 * a0], o = y[2 * a0 + 1];
    out[a0] = a0 == 2 ? m::add_mj(e, o) : m::add(e, o);
    out[a0 + 4] = a0 == 2 ? m::sub_mj(e, o) : m::sub(e, o);
  }
}

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
//   ABL      measurement ablations: 0 none; 1 stream only (no FFT); 2 compute only (no re-load);
//            3 butterflies only (no re-load, no LDS exchange)
//   FULL     every frame brings all N samples (L == N): no zero-padding mask
//   PK       packed-f32 butterflies (see M<PK>)
// ---------------------------------------------------------------------------------------------
// OPT flags
enum : int {
  kPair = 2,     // two frames per wave in flight (needs NBUF == 2)
  kSpread = 4,   // next frame's loads issued from inside passes 1 and 2, one per radix-4 group
  kLdsBlk = 32,  // LDS reads as hand-written ds_read_b64 blocks (no ds_read2_b64 merging)
  kTw1C = 64,    // pass-1 twiddles stored compressed (9 instead of 15 complex values)
  kFence = 128,  // sched_barrier after pass 1
  kRows = 256,   // pass 3 limited to the reference channel plan's output rows
  kMulti = 512,  // a workgroup streams through several consecutive epoch groups
  kPrioValu = 1024, // s_setprio 1 through the butterflies of passes 1 and 2 (where the prefetch loads issue)
  kNoClose = 2048,  // measurement ablation: the epoch close only folds and resets the accumulators
  kTrace = 4096,    // measurement aid: s_memtime at epoch-close entry / exit into the ann_out buffer
  kRegBands = 8192, // epoch close forms the band sums from registers (plans with n_row_entries > 0, no spectrum)
  kHannSym = 16384, // periodic Hann folded into pass 1's first butterflies (w[n + N/2] = 1 - w[n]): 8 window registers
  kTw2Early = 32768, // TW2LDS: the first block of pass-2 twiddles is read from LDS before the butterflies that precede its use
  kAlignedBands = 65536, // N = 4096, equal contiguous bands of 64 / 128 / 256 bins (p.aligned_shift): band sums by DPP + one barrier
  kSc16 = 131072,   // samples in HBM are the radio's wire format (two int16 per complex sample, 4 bytes): converted in pass 1
};

template <int R3_, int NBUF_, bool PREFETCH_, bool NT_, bool MAG_, bool WIN_, bool TW2LDS_, int OCC_, int ABL_,
          bool FULL_, bool PK_, int OPT_ = 0>
struct Cfg {
  static constexpr int OPT = OPT_;  // OR of the flags above
  static constexpr int R3 = R3_, NBUF = NBUF_, OCC = OCC_, ABL = ABL_;
  static constexpr bool PREFETCH = PREFETCH_, NT = NT_, MAG = MAG_, WIN = WIN_, TW2LDS = TW2LDS_, FULL = FULL_,
                        PK = PK_;
  static constexpr bool SC16 = (OPT_ & 131072) != 0;   // kSc16
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
  if constexpr ((C::OPT & kFence) != 0) __builtin_amdgcn_sched_barrier(0);
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

// pass 3 + per-bin accumulate: v[j * R3 + d] is bin a + 16 (m_lo J + j) + 256 d
template <class C>
CRN_DEV void ph_pass3_acc(cx (&u)[16], FrameCtx<C> &c) {
  cx v[16];
  ph_pass3<C>(u, v);
#pragma unroll
  for (int i = 0; i < 16; i++) {
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

// One frame: three register passes + two LDS exchanges + per-bin accumulate.  `u` is clobbered.
// With SPREAD the next frame (`nx`, at `soff_next`) is fetched from inside passes 1 and 2.
template <class C, bool SPREAD = false, bool HALF = false>
CRN_DEV void frame_compute(cx (&u)[16], FrameCtx<C> &c, int f, cx (*nx)[16] = nullptr,
                           __amdgpu_buffer_rsrc_t rsrc = __amdgpu_buffer_rsrc_t(), unsigned voff = 0,
                           unsigned soff_next = 0) {
  using G = Geo<C::R3>;
  cx *buf = c.gbuf + (C::NBUF == 2 ? (f & 1) * G::GROUP_CPLX : 0);
  cx v[16];
  if constexpr (SPREAD) {
    static_assert(C::ABL == 0, "ablations use the plain path");
    const int Lrows = C::FULL ? G::N : c.L;
    const SpreadLoads<C::R3, C::NT, C::SC16> h1{*nx, rsrc, voff, soff_next, 0, HALF, Lrows}, h2{*nx, rsrc, voff, soff_next, 1, HALF, Lrows};
    // Waves in passes 1 and 2 (which also issue the next frame's loads) win VALU arbitration
    // against waves in pass 3 / epoch close: measured +1.4 % (76.9 vs 75.8 %); raising pass 1 alone,
    // pass 3 alone or the LDS phases gains nothing.
    constexpr bool PV = (C::OPT & kPrioValu) != 0;
    if constexpr (PV) __builtin_amdgcn_s_setprio(1);
    ph_pass1<C>(u, v, c, h1);
    if constexpr (PV) __builtin_amdgcn_s_setprio(0);
    if constexpr (G::XWAVE && C::NBUF == 1) __syncthreads();
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
    if constexpr ((C::OPT & kRows) != 0 && C::R3 == 16 && !C::MAG) {
      constexpr unsigned MASK = kRefPlanRows;
#pragma unroll
      for (int i = 0; i < 16; i++) v[i] = cx{0.f, 0.f};
      dft16_pruned<C::PK, MASK>(u, v);
#pragma unroll
      for (int i = 0; i < 16; i++)
        if ((MASK >> i) & 1) c.acc[i] = fmaf(v[i].y, v[i].y, fmaf(v[i].x, v[i].x, c.acc[i]));
      return;
    }
    ph_pass3_acc<C>(u, c);
    return;
  }
  ph_pass1<C>(u, v, c);
  if constexpr (C::ABL == 3) {
#pragma unroll
    for (int i = 0; i < 16; i++) u[i] = v[i];
    ph_pass2<C>(u, v, c);
#pragma unroll
    for (int i = 0; i < 16; i++) u[i] = v[i];
  } else {
    if constexpr (G::XWAVE && C::NBUF == 1) __syncthreads();  // rows may still be read as exchange 2
    ph_x1_write<C>(v, buf, c);
    group_sync<C>();
    ph_x1_read<C>(u, buf, c);
    ph_pass2<C>(u, v, c);
    wave_sync();
    ph_x2_write<C>(v, buf, c);
    wave_sync();
    ph_x2_read<C>(u, buf, c);
  }
  ph_pass3_acc<C>(u, c);
}

// Two frames of the same epoch in one instruction stream, each with its own LDS buffer: the
// LDS writes / reads of one frame are in flight while the butterflies of the other issue, and the
// pair shares its s_barriers (one per frame instead of two).  Needs NBUF == 2.
template <class C>
CRN_DEV void frame_pair_compute(cx (&ua)[16], cx (&ub)[16], FrameCtx<C> &c) {
  using G = Geo<C::R3>;
  static_assert(C::NBUF == 2, "the frame pair uses one exchange buffer per frame");
  cx *bufa = c.gbuf, *bufb = c.gbuf + G::GROUP_CPLX;
  cx va[16], vb[16];
  ph_pass1<C>(ua, va, c);
  group_sync<C>();               // every wave is done reading both buffers (previous pair)
  ph_x1_write<C>(va, bufa, c);
  ph_pass1<C>(ub, vb, c);        // butterflies of B while A's writes drain
  ph_x1_write<C>(vb, bufb, c);
  group_sync<C>();
  ph_x1_read<C>(ua, bufa, c);
  ph_x1_read<C>(ub, bufb, c);
  ph_pass2<C>(ua, va, c);        // B's reads land meanwhile
  wave_sync();
  ph_x2_write<C>(va, bufa, c);
  ph_pass2<C>(ub, vb, c);
  ph_x2_write<C>(vb, bufb, c);
  wave_sync();
  ph_x2_read<C>(ua, bufa, c);
  ph_x2_read<C>(ub, bufb, c);
  ph_pass3_acc<C>(ua, c);
  ph_pass3_acc<C>(ub, c);
}

template <class C>
CRN_DEV void frame_step(cx (&cur)[16], FrameCtx<C> &c, int f, const cx (&u0)[16]) {
  if constexpr (C::ABL >= 2) {
#pragma unroll
    for (int r = 0; r < 16; r++) cur[r] = cx{u0[r].x + (float)f * 1e-30f, u0[r].y};
  }
  if constexpr (C::ABL == 1) {
#pragma unroll
    for (int i = 0; i < 16; i++) c.acc[i] += cur[i].x + cur[i].y;
  } else {
    frame_compute<C>(cur, c, f);
  }
}

// ---------------------------------------------------------------------------------------------
// Epoch close (reference .cpp:157-261 + the reset at :287-288): K-frame averages -> LDS in natural
// bin order -> band sums -> features -> decision.  Resets the accumulators for the next epoch.
// ---------------------------------------------------------------------------------------------
// LDS behind the exchange buffers and the tw2 table, used by the epoch close: the band table copy,
// then [8 teams][16] per-team band partials of the register path.
constexpr int kCloseLdsBytes = kBandTabWords * 4 + 8 * 16 * 4;

// LDS address-space views for the epoch close (see epoch_close): ds_* instructions, lgkmcnt only.
typedef __attribute__((address_space(3))) float lds_f32;
typedef __attribute__((address_space(3))) int lds_i32;
typedef __attribute__((address_space(3))) double lds_f64;
CRN_DEV unsigned lds_offset(const void *p) {
  return (unsigned)(unsigned long long)(__attribute__((address_space(3))) const void *)p;
}

// Sum over a team of TEAM consecutive lanes (32 or 64), every lane gets the total: butterflies on
// the DPP path (quad_perm xor 1, xor 2, row_half_mirror, row_mirror) up to 16-lane rows, then the
// four row sums come back through v_readlane.  No LDS-pipe shuffles.
template <int CTRL>
CRN_DEV float dpp_add(float v) {
  return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
template <int TEAM>
CRN_DEV float team_sum(float v, int tid) {
  static_assert(TEAM == 32 || TEAM == 64, "team is half a wave or a wave");
  v = dpp_add<0xB1>(v);   // quad_perm [1,0,3,2]
  v = dpp_add<0x4E>(v);   // quad_perm [2,3,0,1]
  v = dpp_add<0x141>(v);  // row_half_mirror
  v = dpp_add<0x140>(v);  // row_mirror
  const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
  const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
  const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
  const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
  if constexpr (TEAM == 64) return (r0 + r1) + (r2 + r3);
  else return (tid & 32) ? r2 + r3 : r0 + r1;
}

// W consecutive table words through the scalar cache into SGPRs (the epoch close must not touch
// vmcnt, and an LDS read queues behind the other waves' exchange traffic): issue with s_load_row,
// then s_wait_row before the first use.
template <int W> struct SWords;
template <> struct SWords<2> { typedef int T __attribute__((ext_vector_type(2))); };
template <> struct SWords<4> { typedef int T __attribute__((ext_vector_type(4))); };
template <> struct SWords<8> { typedef int T __attribute__((ext_vector_type(8))); };
template <> struct SWords<16> { typedef int T __attribute__((ext_vector_type(16))); };
template <int W, int BYTE_OFF>
CRN_DEV typename SWords<W>::T s_load_row(const int *base) {
  typename SWords<W>::T r;
  if constexpr (W == 2) asm volatile("s_load_dwordx2 %0, %1, %2" : "=s"(r) : "s"(base), "n"(BYTE_OFF));
  if constexpr (W == 4) asm volatile("s_load_dwordx4 %0, %1, %2" : "=s"(r) : "s"(base), "n"(BYTE_OFF));
  if constexpr (W == 8) asm volatile("s_load_dwordx8 %0, %1, %2" : "=s"(r) : "s"(base), "n"(BYTE_OFF));
  if constexpr (W == 16) asm volatile("s_load_dwordx16 %0, %1, %2" : "=s"(r) : "s"(base), "n"(BYTE_OFF));
  return r;
}
template <class V>
CRN_DEV void s_wait_row(V &r) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(r));
}

// The reference's 4-5-3 sigmoid network and cascade (CE_Predictive_Node.cpp:200-261), spread over the
// lanes of a team: hidden unit j on lane j, output k on lane k, values passed with v_readlane.  Each
// unit's sum is formed by one lane in the reference's order, so the results are those of the serial
// loop; what changes is the latency — two exp() in sequence instead of eight (one lane doing all of
// it cost the reference-mode kernel 3.8 %).  Weights come from the LDS copy of the table (a per-lane
// global load would wait on vmcnt behind the prefetch).  Every lane of the team must call this.
template <int TEAM>
CRN_DEV double lane_f64(double v, int src, int half) {
  const int lo = __builtin_amdgcn_readlane((int)(__double_as_longlong(v) & 0xffffffffll), src);
  const int hi = __builtin_amdgcn_readlane((int)(__double_as_longlong(v) >> 32), src);
  double r = __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
  if constexpr (TEAM == 32) {  // two groups share a wave: the upper one reads lanes 32 + src
    const int lo2 = __builtin_amdgcn_readlane((int)(__double_as_longlong(v) & 0xffffffffll), 32 + src);
    const int hi2 = __builtin_amdgcn_readlane((int)(__double_as_longlong(v) >> 32), 32 + src);
    const double r2 = __longlong_as_double(((long long)hi2 << 32) | (unsigned int)lo2);
    r = half ? r2 : r;
  }
  return r;
}

template <int TEAM>
CRN_DEV void ann_decide_team(const SenseParams &p, const lds_f64 *w_ih, const lds_f64 *w_ho, long long epoch,
                             bool store, int lane, int half, float nf, float ch1, float ch2, float ch3) {
  // .cpp:200: Features_Buffer = {0, NOISE_FLOOR, CH1, CH2, CH3} widened to double
  const double f1 = (double)nf, f2 = (double)ch1, f3 = (double)ch2, f4 = (double)ch3;
  const int j = (lane >= 1 && lane <= 5) ? lane : 1;  // .cpp:214-220, unit j
  double s = w_ih[0 * 6 + j];
  s += f1 * w_ih[1 * 6 + j];
  s += f2 * w_ih[2 * 6 + j];
  s += f3 * w_ih[3 * 6 + j];
  s += f4 * w_ih[4 * 6 + j];
  const double hj = 1.0 / (1.0 + exp(-s));
  double hid[6];
#pragma unroll
  for (int q = 1; q <= 5; q++) hid[q] = lane_f64<TEAM>(hj, q, half);
  const int k = (lane >= 1 && lane <= 3) ? lane : 1;  // .cpp:229-235, output k
  double so = w_ho[0 * 4 + k];
#pragma unroll
  for (int q = 1; q <= 5; q++) so += hid[q] * w_ho[q * 4 + k];
  const double ok = 1.0 / (1.0 + exp(-so));
  const double o1 = lane_f64<TEAM>(ok, 1, half), o2 = lane_f64<TEAM>(ok, 2, half), o3 = lane_f64<TEAM>(ok, 3, half);
  // .cpp:245-261 cascade
  int d = 0;
  if (o1 >= p.ann_threshold) d = 1;
  else if (o2 >= p.ann_threshold) d = 2;
  else if (o3 >= p.ann_threshold) d = 3;
  if (store) {
    if (lane >= 1 && lane <= 3 && p.ann_out != nullptr) p.ann_out[epoch * 3 + (lane - 1)] = ok;
    if (lane == 0 && p.decision != nullptr) p.decision[epoch] = d;
    if (lane < p.n_bands && p.occupancy != nullptr) p.occupancy[epoch * p.n_bands + lane] = (uint8_t)(lane >= 1 && lane == d);
  }
}

template <class C>
CRN_DEV void epoch_close(FrameCtx<C> &c, const SenseParams &p, long long epoch_base) {
  constexpr int R3 = C::R3;
  constexpr bool MAG = C::MAG;
  using G = Geo<R3>;
  constexpr int T = G::T, N = G::N, J = G::J;
  float (&acc)[16] = c.acc;
  const float Kf = c.Kf;
  // Everything this block needs is re-derived here from uniform values (SGPRs) and the hardware
  // lane id, so that nothing but the accumulators stays live in VGPRs across the frame loop for a
  // block that runs once per K frames: what the allocator kept for it, it spilled, and a scratch
  // reload waits on vmcnt behind the next frame's prefetch.
  if constexpr ((C::OPT & kNoClose) != 0) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; i++) {
      s += acc[i];
      acc[i] = 0.f;
    }
    if (s == 123.456f && p.features != nullptr) p.features[0] = s;  // keeps the accumulation live
    if constexpr ((C::OPT & kTrace) != 0 && G::XWAVE) __syncthreads();  // variant 18: what one barrier per epoch costs
    return;
  }
  // latency-bound stretch with nothing of this wave's in flight behind it, and the workgroup's other
  // waves waiting at its barriers: outrank the butterflies (+0.6 % at N = 4096, +1.3 % at 2048; the
  // barrier-free sizes lose 0.5 % with it)
  if constexpr ((C::OPT & kPrioValu) != 0 && G::XWAVE) __builtin_amdgcn_s_setprio(3);
  const int tid = c.wave * 64 + (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
  const int t = tid % T, grp = tid / T;
  const long long epoch = epoch_base + (long long)grp * c.grp_epoch_stride;
  const bool active = epoch < p.n_epochs;
  const int a = t / R3, m_lo = t % R3;
  [[maybe_unused]] unsigned long long tr0 = 0, tr1 = 0, tr2 = 0, tr3 = 0;
  if constexpr ((C::OPT & kTrace) != 0) {
    // [epoch][3] uint64: entry of the group's first wave; its later stamps as four 16-bit deltas
    // (band sums done, barrier passed, features ready, exit); entry of the group's last wave
    unsigned long long *tr = reinterpret_cast<unsigned long long *>(p.ann_out);
    // entries in s_memrealtime (100 MHz, the same clock on every XCD: comparable across workgroups);
    // the deltas inside the block in s_memtime (shader clock, finer, per-XCD)
    const unsigned long long wall = __builtin_amdgcn_s_memrealtime();
    tr0 = __builtin_amdgcn_s_memtime();
    if (active && tr != nullptr) {
      if (t == 0) tr[epoch * 3 + 0] = wall;
      if (t == T - 1) tr[epoch * 3 + 2] = wall;
    }
  }
  // Energy mode: the division by K is applied to the band sums (and to the per-bin values only
  // when a spectrum is requested) — sixteen IEEE divides per thread per epoch were half a frame's
  // worth of VALU work.  Sum-then-divide differs from the reference order (divide-then-sum) by
  // rounding only.
  // The opaque values are 32-bit LDS offsets, not generic pointers: through a generic pointer every
  // access below became a FLAT instruction followed by s_waitcnt vmcnt(0), which also drained the
  // next frame's prefetch at every epoch close.
  const unsigned tab_off = c.lds_base + (unsigned)((G::GROUPS * C::NBUF * G::GROUP_CPLX + 16 * R3) * sizeof(cx));
  const unsigned gb_off = c.lds_base + (unsigned)grp * (unsigned)(C::NBUF * G::GROUP_CPLX * sizeof(cx));
  const lds_i32 *tab = reinterpret_cast<const lds_i32 *>(tab_off);
  const lds_f32 *thr = reinterpret_cast<const lds_f32 *>(tab_off + 416 * 4);
  const lds_f64 *w_ih = reinterpret_cast<const lds_f64 *>(tab_off + 544 * 4);  // [5][6]
  const lds_f64 *w_ho = reinterpret_cast<const lds_f64 *>(tab_off + 604 * 4);  // [6][4]
  lds_f32 *spec = reinterpret_cast<lds_f32 *>(gb_off);    // N + N/16 floats
  lds_f32 *featl = spec + spec_phys(N);                   // CRN_MAX_BANDS floats (LDS path)
  constexpr int TEAM = G::TEAM;
  constexpr int TPG = T / TEAM;  // teams (waves) per group
  const int lane = t % TEAM;
  lds_f32 *part = reinterpret_cast<lds_f32 *>(tab_off + kBandTabWords * 4);  // [256 / TEAM][16]
  [[maybe_unused]] const lds_f32 *feat = nullptr;
  // Three forms of the close, chosen per launch (one kernel holding several spilled in the frame loop):
  if constexpr ((C::OPT & kAlignedBands) != 0) {
    // Equal contiguous bands of W = 2^sh bins, sh = 6..8 (the Welch scan's 64 channels of 64 bins), N = 4096:
    // thread (a, m_lo) holds bins 256 d + 16 m_lo + a in acc[d], so band (256 d + 16 m_lo) >> sh is the sum over
    // all 16 a and over a group of G = W / 16 consecutive m_lo — lanes of one DPP row.  Group sums by DPP
    // (no LDS), the four rows of a wave through the wave's OWN exchange rows (only x1 writes of other waves
    // ever touch them, and those sit between the frame's two barriers), one barrier, then one lane per band adds
    // the 16 values of a: 2-4 DPP adds per register, <= 16 narrow LDS writes and one barrier instead of a
    // spectrum image, three barriers and a table walk (5 % of the Welch stream at K = 8).
    static_assert(R3 == 16 && !MAG, "aligned-band close: N = 4096, energy mode");
    const int sh = p.aligned_shift;   // uniform
    const int G = 1 << (sh - 4), nb = p.n_bands, al = (tid >> 4) & 3, r = m_lo & (G - 1), grp_b = m_lo >> (sh - 4);
    constexpr int kStride = 72;       // floats per a-row of partials: 72 mod 32 = 8 keeps a wave's rows on distinct banks
    lds_f32 *mine = reinterpret_cast<lds_f32 *>(c.lds_base + (unsigned)(4 * c.wave * G::ROW * sizeof(cx))) + al * kStride;
    const float thr_lane = thr[tid & 63];
#pragma unroll
    for (int d = 0; d < 16; d++) {
      float v = acc[d];
      acc[d] = 0.f;  // .cpp:287
      v = dpp_add<0xB1>(v);
      v = dpp_add<0x4E>(v);
      if (sh >= 7) v = dpp_add<0x141>(v);
      if (sh >= 8) v = dpp_add<0x140>(v);
      if ((d & (G - 1)) == r) mine[(d << (8 - sh)) + grp_b] = v;   // one lane of the group stores the group's sum
    }
    if constexpr ((C::OPT & kTrace) != 0) tr1 = __builtin_amdgcn_s_memtime();
    __syncthreads();
    if constexpr ((C::OPT & kTrace) != 0) tr2 = __builtin_amdgcn_s_memtime();
    if (c.wave == 0) {
      const int b = tid;  // one lane per band (n_bands <= 64)
      float sum = 0.f;
      if (b < nb) {
#pragma unroll
        for (int w4 = 0; w4 < 4; w4++) {
          const lds_f32 *src = reinterpret_cast<const lds_f32 *>(c.lds_base + (unsigned)(4 * w4 * G::ROW * sizeof(cx)));
          const float a0 = src[b], a1 = src[kStride + b], a2 = src[2 * kStride + b], a3 = src[3 * kStride + b];
          sum += a0;
          sum += a1;
          sum += a2;
          sum += a3;
        }
      }
      if constexpr ((C::OPT & kTrace) != 0) tr3 = __builtin_amdgcn_s_memtime();
      const float f = sc_unscale<C>(__fdiv_rn(sum, Kf), p);
      const bool in = b < nb;
      if (p.decide == CRN_DECIDE_THRESHOLD_K) {
        const float ref = p.ref_band >= 0 ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(f), p.ref_band)) : 1.0f;
        const bool occ = active && in && f > thr_lane * ref;
        if (active && in && p.occupancy != nullptr) p.occupancy[epoch * nb + b] = (uint8_t)occ;
        const unsigned long long m = __ballot(occ);
        if (active && b == 0 && p.decision != nullptr) p.decision[epoch] = __popcll(m);
      } else if (active) {
        if (b == 0 && p.decision != nullptr) p.decision[epoch] = 0;
        if (in && p.occupancy != nullptr) p.occupancy[epoch * nb + b] = 0;
      }
      if (active && in && p.features != nullptr) p.features[epoch * nb + b] = f;
    }
  } else if constexpr ((C::OPT & kRegBands) != 0) {
    // Band sums straight from the accumulator registers: no LDS image of the spectrum, no barrier
    // before it (nothing aliases the exchange buffers) and none after the decision.  Thread bins are
    // base_j + 256 d; the host cut the band plan at the 256-bin rows (crn_api.cpp), so each entry is
    // (row d, band, [lo, hi) in the row): masked add over j, DPP team sum, lane `band` keeps it.
    // LDS round trips are what this block avoids (an LDS read here queues behind the exchange
    // traffic of the CU's other waves: measured ~500 ticks each): the entries come through the
    // scalar cache, the features stay in lanes, the thresholds are fetched first and used last.
    constexpr int CAP = kRowEntryWords / R3;  // entry slots per row
    constexpr auto row_live = [](int d) {
      return !((C::OPT & kRows) != 0 && R3 == 16 && !MAG) || ((kRefPlanRows >> d) & 1) != 0;
    };
    float thr_lane = 0.f;
    if constexpr (TPG == 1) thr_lane = thr[lane & 15];  // fetched first, used last
    // one row's entries at a time, the next row's load in flight meanwhile: holding all of them
    // costs SGPRs the frame loop needs (the spill lanes' VGPR pushed a loop address to scratch)
    constexpr auto next_live = [](int d) {
      for (int x = d + 1; x < R3; x++)
        if (!((C::OPT & kRows) != 0 && R3 == 16 && !MAG) || ((kRefPlanRows >> x) & 1) != 0) return x;
      return (int)R3;
    };
    constexpr int kFirst = next_live(-1);
    float fsum = 0.f;
    typename SWords<CAP>::T ent_next = s_load_row<CAP, (512 + kFirst * CAP) * 4>(p.band_tab);
    static_for<R3>([&](auto dc) {
      constexpr int d = decltype(dc)::value;
      if constexpr (row_live(d)) {
        typename SWords<CAP>::T ent = ent_next;
        s_wait_row(ent);
        constexpr int dn = next_live(d);
        if constexpr (dn < R3) ent_next = s_load_row<CAP, (512 + dn * CAP) * 4>(p.band_tab);
#pragma unroll
        for (int e = 0; e < CAP; e++) {
          const int w = ent[e];
          if (w != 0) {  // uniform; 0 = unused slot
            const int band = w >> 18, lo = (w >> 9) & 511, span = (w & 511) - lo;
            float v = 0.f;
#pragma unroll
            for (int j = 0; j < J; j++) {
              const int base = a + 16 * (m_lo * J + j);
              v += (unsigned)(base - lo) < (unsigned)span ? acc[j * R3 + d] : 0.f;
            }
            v = team_sum<TEAM>(v, tid);
            fsum += lane == band ? v : 0.f;
          }
        }
      }
    });
#pragma unroll
    for (int i = 0; i < 16; i++) acc[i] = 0.f;  // .cpp:287
    if constexpr ((C::OPT & kTrace) != 0) tr1 = __builtin_amdgcn_s_memtime();
    if constexpr (TPG > 1) {
      if (lane < 16) part[(tid / TEAM) * 16 + lane] = fsum;
      __syncthreads();
      if constexpr ((C::OPT & kTrace) != 0) tr2 = __builtin_amdgcn_s_memtime();
      if (t < TEAM && lane < 16) {
        thr_lane = thr[lane];  // same LDS round trip as the partials
        fsum = 0.f;
#pragma unroll
        for (int w = 0; w < TPG; w++) fsum += part[(grp * TPG + w) * 16 + lane];
      }
    }
    if constexpr ((C::OPT & kTrace) != 0) tr3 = __builtin_amdgcn_s_memtime();
    // the first team of the group stores and decides; lane b holds band b (n_bands <= 16)
    if (t < TEAM) {
      const float fs1 = MAG ? sc_unscale<C>(fsum, p) : fsum;
      const float f = MAG ? fs1 * fs1 : sc_unscale<C>(__fdiv_rn(fsum, Kf), p);  // .cpp:194-197
      const int half = TEAM == 32 ? (tid & 32) : 0;             // two groups share a wave at T = 32
      auto from_lane = [&](int b) {
        const float lo_half = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(f), b));
        if constexpr (TEAM == 32) {
          const float hi_half = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(f), 32 + b));
          return half ? hi_half : lo_half;
        }
        return lo_half;
      };
      const bool in = lane < p.n_bands;
      if (p.decide == CRN_DECIDE_ANN_K) {
        const float nf = from_lane(0), ch1 = from_lane(1), ch2 = from_lane(2), ch3 = from_lane(3);
        ann_decide_team<TEAM>(p, w_ih, w_ho, epoch, active, lane, half, nf, ch1, ch2, ch3);
      } else if (p.decide == CRN_DECIDE_THRESHOLD_K) {
        const float ref = p.ref_band >= 0 ? from_lane(p.ref_band) : 1.0f;
        const bool occ = active && in && f > thr_lane * ref;
        if (active && in && p.occupancy != nullptr) p.occupancy[epoch * p.n_bands + lane] = (uint8_t)occ;
        unsigned long long m = __ballot(occ);
        if constexpr (TEAM == 32) m = (m >> half) & 0xffffffffull;
        if (active && t == 0 && p.decision != nullptr) p.decision[epoch] = __popcll(m);
      } else if (active) {
        if (t == 0 && p.decision != nullptr) p.decision[epoch] = 0;
        if (in && p.occupancy != nullptr) p.occupancy[epoch * p.n_bands + lane] = 0;
      }
      if (active && in && p.features != nullptr) p.features[epoch * p.n_bands + lane] = f;
    }
  } else {
    // descriptor of this lane's first band (one lane per band below): fetched now, used after the barriers
    int pre_s0 = 0, pre_s1 = 0, pre_lo = 0, pre_hi = 0;
    if (t < p.n_bands) {
      pre_s0 = tab[t];
      pre_s1 = tab[t + 1];
      if (pre_s1 > pre_s0) {
        pre_lo = tab[96 + pre_s0];
        pre_hi = tab[256 + pre_s0];
      }
    }
    if constexpr (G::XWAVE) __syncthreads();
    else wave_sync();
#pragma unroll
    for (int j = 0; j < J; j++)
#pragma unroll
      for (int d = 0; d < R3; d++) {
        // the row-pruned kernel never accumulates (or reads back) the other rows
        if constexpr ((C::OPT & kRows) != 0 && R3 == 16 && !MAG) {
          if (!((kRefPlanRows >> d) & 1)) continue;
        }
        const int k = a + 16 * (m_lo * J + j) + 256 * d;
        spec[spec_phys(k)] = acc[j * R3 + d];
      }
#pragma unroll
    for (int i = 0; i < 16; i++) acc[i] = 0.f;  // .cpp:287
    if constexpr (G::XWAVE) __syncthreads();
    else wave_sync();

    if constexpr ((C::OPT & kTrace) != 0) tr1 = __builtin_amdgcn_s_memtime();  // LDS form: spectrum image visible
    if (p.spectrum != nullptr && active) {
      float *dst = p.spectrum + epoch * N;
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const float x = spec[spec_phys(t + T * r)];
        dst[t + T * r] = sc_unscale<C>(MAG ? x : __fdiv_rn(x, Kf), p);
      }
    }

    // band sums (reference .cpp:173-191).  This stretch is pure latency (the wave has no loads in
    // flight beyond its prefetched frame), every LDS round trip queues behind the CU's exchange
    // traffic, and a dependent VALU chain issues one instruction per ~10 cycles — so the work is
    // spread over lanes instead of bands (one team per band, 16 bands in sequence per wave, cost
    // the 64-band Welch kernel 11 us per epoch, 40 % of the wave's time):
    //   every thread sums the 16 consecutive bins of "its" block (thread t: bins 16 t .. 16 t + 15,
    //   ascending like the reference) -> blk[t]; the 16 lanes of a DPP row add their block totals
    //   -> rows[t / 16] (256 bins);  then ONE LANE PER BAND walks its segments in steps of 256, 16
    //   and 1 bins (at most 15 + 15 + 16 + 15 + 15 reads), all bands at once.
    lds_f32 *blk = featl + 80;  // [T] block totals (behind the CRN_MAX_BANDS features)
    lds_f32 *rows = blk + T;                 // [R3] row totals
    {
      float bs = 0.f;
#pragma unroll
      for (int j = 0; j < 16; j++) bs += spec[17 * t + j];  // spec_phys(16 t + j)
      blk[t] = bs;
      float rs = dpp_add<0xB1>(bs);   // the 16 block totals of a 256-bin row sit in one DPP row
      rs = dpp_add<0x4E>(rs);
      rs = dpp_add<0x141>(rs);
      rs = dpp_add<0x140>(rs);
      if ((t & 15) == 0) rows[t >> 4] = rs;
    }
    if constexpr (G::XWAVE) __syncthreads();
    else wave_sync();
    // four block (or row) totals per LDS round trip, added in ascending order
    auto add_run = [&](float &sum, const lds_f32 *tot, int i, int n) {  // tot[i .. i + n)
      for (; n >= 4; n -= 4, i += 4) {
        const float a0 = tot[i], a1 = tot[i + 1], a2 = tot[i + 2], a3 = tot[i + 3];
        sum += a0;
        sum += a1;
        sum += a2;
        sum += a3;
      }
      for (; n > 0; n--, i++) sum += tot[i];
    };
    for (int b = t; b < p.n_bands; b += T) {
      float sum = 0.f;
      const bool first = b == t;  // this lane's first band: descriptor fetched before the barriers
      const int s0 = first ? pre_s0 : tab[b], s1 = first ? pre_s1 : tab[b + 1];
      for (int sg = s0; sg < s1; sg++) {
        int k = (first && sg == s0) ? pre_lo : tab[96 + sg];
        const int hi = (first && sg == s0) ? pre_hi : tab[256 + sg];
        while (k < hi && (k & 15) != 0) sum += spec[spec_phys(k++)];
        if (k + 16 <= hi) {
          // whole blocks up to the next row boundary, whole rows, whole blocks after them
          int n = ((hi - k) >> 4);                       // whole blocks available
          const int to_row = ((256 - (k & 255)) & 255) >> 4;  // blocks until k is row-aligned
          const int head = n < to_row ? n : to_row;
          add_run(sum, blk, k >> 4, head);
          k += head * 16;
          n -= head;
          const int nrows = n >> 4;
          add_run(sum, rows, k >> 8, nrows);
          k += nrows * 256;
          n -= nrows * 16;
          add_run(sum, blk, k >> 4, n);
          k += n * 16;
        }
        while (k < hi) sum += spec[spec_phys(k++)];
      }
      const float msum = MAG ? sc_unscale<C>(sum, p) : sum;
      featl[b] = MAG ? msum * msum : sc_unscale<C>(__fdiv_rn(sum, Kf), p);  // .cpp:194-197
    }
    if constexpr ((C::OPT & kTrace) != 0) tr2 = __builtin_amdgcn_s_memtime();  // LDS form: this wave's band sums done
    if constexpr (G::XWAVE) __syncthreads();
    else wave_sync();
    if constexpr ((C::OPT & kTrace) != 0) tr3 = __builtin_amdgcn_s_memtime();  // LDS form: every feature written

    feat = featl;

    // LDS path: the first team of the group stores and decides.
    if (active && t < TEAM) {
      if (p.features != nullptr)
        for (int b = t; b < p.n_bands; b += TEAM) p.features[epoch * p.n_bands + b] = feat[b];

      if (p.decide == CRN_DECIDE_ANN_K) {
        ann_decide_team<TEAM>(p, w_ih, w_ho, epoch, true, lane, TEAM == 32 ? (tid & 32) : 0, feat[0], feat[1], feat[2], feat[3]);
      } else if (p.decide == CRN_DECIDE_THRESHOLD_K) {
        // lane i takes bands i, i + TEAM, ...; the count of occupied bands is a ballot, not a serial walk
        {
          const float ref = p.ref_band >= 0 ? feat[p.ref_band] : 1.0f;
          int cnt = 0;
          for (int b0 = 0; b0 < p.n_bands; b0 += TEAM) {
            const int b = b0 + t;
            const bool in = b < p.n_bands;
            const bool occ = in && feat[in ? b : 0] > thr[in ? b : 0] * ref;
            if (in && p.occupancy != nullptr) p.occupancy[epoch * p.n_bands + b] = (uint8_t)occ;
            unsigned long long m = __ballot(occ);
            if constexpr (TEAM == 32) m = (m >> (tid & 32)) & 0xffffffffull;
            cnt += __popcll(m);
          }
          if (t == 0 && p.decision != nullptr) p.decision[epoch] = cnt;
        }
      } else {
        if (t == 0 && p.decision != nullptr) p.decision[epoch] = 0;
        if (p.occupancy != nullptr)
          for (int b = t; b < p.n_bands; b += TEAM) p.occupancy[epoch * p.n_bands + b] = 0;
      }
    }
  }
  // With one exchange buffer the next epoch's first frame syncs the workgroup before it writes
  // exchange 1 (frame_compute), which is after every wave has passed this point: no barrier here.
  if constexpr (G::XWAVE && C::NBUF == 2) __syncthreads();
  if constexpr (!G::XWAVE) wave_sync();
  if constexpr ((C::OPT & kTrace) != 0) {
    unsigned long long *tr = reinterpret_cast<unsigned long long *>(p.ann_out);
    const unsigned long long tr4 = __builtin_amdgcn_s_memtime();
    auto d16 = [&](unsigned long long x) { return (x - tr0) > 0xFFFFull ? 0xFFFFull : (x - tr0); };
    if (active && tr != nullptr && t == 0)
      tr[epoch * 3 + 1] = d16(tr1) | (d16(tr2) << 16) | (d16(tr3) << 32) | (d16(tr4) << 48);
  }
  if constexpr ((C::OPT & kPrioValu) != 0 && G::XWAVE) __builtin_amdgcn_s_setprio(0);
}

// Buffer resource over the IQ window of epoch group `eg` (GROUPS consecutive epochs): anything
// past the window, or past the end of the batch, reads as zero.
template <int R3, int SB = 8>
CRN_DEV __amdgpu_buffer_rsrc_t group_rsrc(const SenseParams &p, long long eg, int span = 1) {
  using G = Geo<R3>;
  const long long first = eg * G::GROUPS * p.epoch_stride;
  long long left = (p.total_samples - first) * SB;
  const long long window = ((long long)span * G::GROUPS * p.epoch_stride + (long long)p.K * p.frame_stride + G::N) * SB;
  if (left > window) left = window;
  if (left < 0) left = 0;
  char *base = reinterpret_cast<char *>(const_cast<float2 *>(p.iq)) + first * SB;   // p.iq is int16 pairs when SB == 4
  return __builtin_amdgcn_make_buffer_rsrc(base, 0, (int)left, 0x00020000);
}

// Epoch groups [g0, g0 + n_local) of a streaming workgroup (graded launch: launch_cfg).
struct StreamSpan {
  long long g0;
  int epw, n_local;
};
template <int R3>
CRN_DEV StreamSpan stream_span(const SenseParams &p) {
  using G = Geo<R3>;
  const long long n_groups = (p.n_epochs + G::GROUPS - 1) / G::GROUPS;
  const bool big = (long long)blockIdx.x < p.n_big_wgs;
  StreamSpan s;
  s.epw = big ? p.groups_per_wg : 1;
  s.g0 = big ? (long long)blockIdx.x * p.groups_per_wg
             : p.n_big_wgs * p.groups_per_wg + ((long long)blockIdx.x - p.n_big_wgs);
  s.n_local = (int)((n_groups - s.g0) < s.epw ? (n_groups - s.g0) : s.epw);
  return s;
}

// ---------------------------------------------------------------------------------------------
// the sensing kernel
// ---------------------------------------------------------------------------------------------
template <class C>
__global__ __launch_bounds__(256, C::OCC) void sense_kernel(const SenseParams p) {
  constexpr int R3 = C::R3, NBUF = C::NBUF;
  constexpr bool NT = C::NT, SC = C::SC16;
  constexpr unsigned SB = C::SB;
  using G = Geo<R3>;
  constexpr int T = G::T;
  extern __shared__ __attribute__((aligned(16))) cx lds[];

  const int tid = threadIdx.x;
  const int grp = tid / T;
  const int t = tid % T;       // pass-1 column, n_lo
  const int a = t / R3;        // pass-2/3 sub-transform id (k mod 16)
  const int m_lo = t % R3;     // pass-2 column / pass-3 slot g

  FrameCtx<C> c;
  c.t = t;
  c.a = a;
  c.m_lo = m_lo;
  c.L = p.L;
  c.gbuf = lds + grp * (NBUF * G::GROUP_CPLX);
  c.wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  c.grp_epoch_stride = 1;
  c.lds_base = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_offset(lds));
  c.tw2_lds = lds + G::GROUPS * NBUF * G::GROUP_CPLX;  // [16][R3], TW2LDS only
  const int K = p.K;
  c.Kf = (float)K;
  c.invK = 1.0f / (float)K;

  // frame-invariant twiddles, kept in registers across frames and epochs
#pragma unroll
  for (int i = 1; i < ((C::OPT & kTw1C) != 0 ? 9 : 16); i++) c.tw1[i] = reinterpret_cast<const cx *>(p.tw1)[i * T + t];
  if constexpr ((C::OPT & kTw1C) != 0) c.tw1[0] = reinterpret_cast<const cx *>(p.tw1)[16 * T + t];  // W_N^{16 t}
  {
    // band table -> LDS (2 KiB behind the exchange buffers and the tw2 table): the epoch close walks
    // it, and from global memory every walk step was a dependent ~1 us vector load
    int *tab = reinterpret_cast<int *>(lds + G::GROUPS * NBUF * G::GROUP_CPLX + 16 * R3);
    tab[tid] = p.band_tab[tid];
    tab[tid + 256] = p.band_tab[tid + 256];
    if (tid < kBandTabWords - 512) tab[tid + 512] = p.band_tab[tid + 512];  // row entries
  }
  if constexpr (C::TW2LDS) {
    if (tid < 16 * R3) lds[G::GROUPS * NBUF * G::GROUP_CPLX + tid] = reinterpret_cast<const cx *>(p.tw2)[tid];
  }
  __syncthreads();
  if constexpr (!C::TW2LDS) {
#pragma unroll
    for (int i = 1; i < 16; i++) c.tw2[i] = reinterpret_cast<const cx *>(p.tw2)[i * R3 + m_lo];
  }
  if constexpr (C::WIN && (C::OPT & kHannSym) != 0) {
#pragma unroll
    for (int q = 0; q < 4; q++) c.winp[q] = cx{p.window[t + T * (2 * q)], p.window[t + T * (2 * q + 1)]};
  } else if constexpr (C::WIN) {
#pragma unroll
    for (int r = 0; r < 16; r++) c.win[r] = p.window[t + T * r];
  }
#pragma unroll
  for (int i = 0; i < 16; i++) c.acc[i] = 0.f;

  constexpr unsigned kNowhere = 0x80000000u;  // scalar offset past every window: dropped by the range check
  const unsigned voff = (unsigned)(grp * (unsigned)p.epoch_stride + t) * SB;
  const unsigned fbytes = (unsigned)p.frame_stride * SB;

  cx ua[16], ub[16];
  [[maybe_unused]] cx u0[16];

  if constexpr (C::WIN && C::ABL == 0 && C::PREFETCH && (C::OPT & kMulti) != 0) {
    if (p.frame_stride * 2 == G::N && p.epoch_stride == (long long)K * (G::N / 2)) {
      // Welch (hop = N/2) over dense epochs: a lane group's epochs are one uninterrupted stream of
      // half-frames H(g) = samples [g N/2, (g+1) N/2) — frame g = H(g) | H(g+1), and the half an epoch
      // ends with is the half the next one starts with.  Three half-frame register sets: two hold the
      // current frame's raw samples, the third receives H(g+2) while frame g is computed, so every
      // sample is fetched once per lane group and the prefetch runs across epoch boundaries; the
      // close fires after every K-th frame.  The workgroup's GROUPS x epw epochs are dealt to its
      // lane groups in runs of epw (group g: epochs E0 + g epw ...), so that each group's stream is
      // contiguous; every group runs epw x K frames (the ragged end closes inactive epochs: uniform
      // barriers, loads past the batch return zero).
      constexpr unsigned hbytes = (unsigned)(G::N / 2) * SB;
      const StreamSpan sp = stream_span<R3>(p);
      const __amdgpu_buffer_rsrc_t rs = group_rsrc<R3, (int)SB>(p, sp.g0, sp.epw);
      const unsigned voff = (unsigned)(grp * sp.epw * (unsigned)p.epoch_stride + t) * SB;  // shadows the per-epoch one
      c.grp_epoch_stride = sp.epw;
      load_frame<R3, NT, SC>(ua, rs, voff, 0u);
      // Three half-frame sets whose roles rotate (current low half, current high half, incoming):
      // the loop is unrolled by three so the rotation is a renaming, not 16 register moves a frame.
      cx ha[16], hb[16], hc[16];  // only [0, 8) of each is used (frame_compute's prefetch target is a cx[16])
#pragma unroll
      for (int r = 0; r < 8; r++) {
        ha[r] = ua[r];
        hb[r] = ua[8 + r];
      }
      const int F = (G::GROUPS == 1 ? sp.n_local : sp.epw) * K;
      int f = 0, j = 0, g = 0;
#define CRN_WELCH_STEP(LO, HI, IN)                                                                  \
      {                                                                                             \
        _Pragma("unroll") for (int r = 0; r < 8; r++) {                                             \
          ub[r] = LO[r];                                                                            \
          ub[8 + r] = HI[r];                                                                        \
        }                                                                                           \
        frame_compute<C, true, true>(ub, c, f, &IN, rs, voff, g + 1 < F ? (unsigned)(g + 2) * hbytes : kNowhere); \
        if (++f == K) {                                                                             \
          f = 0;                                                                                    \
          epoch_close<C>(c, p, sp.g0 * G::GROUPS + j);                                              \
          j++;                                                                                      \
        }                                                                                           \
        g++;                                                                                        \
      }
      while (g < F) {
        CRN_WELCH_STEP(ha, hb, hc)
        if (g >= F) break;
        CRN_WELCH_STEP(hb, hc, ha)
        if (g >= F) break;
        CRN_WELCH_STEP(hc, ha, hb)
      }
#undef CRN_WELCH_STEP
      return;
    }
  }
  {
    const long long epoch_base = (long long)blockIdx.x * G::GROUPS;
    const __amdgpu_buffer_rsrc_t rsrc = group_rsrc<R3, (int)SB>(p, blockIdx.x);
    load_frame<R3, NT, SC>(ua, rsrc, voff, 0u, C::FULL ? G::N : c.L);
    if constexpr (C::ABL >= 2) {
#pragma unroll
      for (int r = 0; r < 16; r++) u0[r] = ua[r];
    }
    if constexpr (C::WIN && C::ABL == 0 && C::PREFETCH) {
      if (p.frame_stride * 2 == G::N) {
        // Welch, hop = N/2: frame f = halves H(f) | H(f+1) with H(j) = samples [j N/2, (j+1) N/2).
        // Three half-frame register sets: two hold the current frame's raw samples, the third
        // receives H(f+2) while frame f is computed, so every sample is fetched from HBM once per
        // epoch.  (ua was loaded as a whole frame above: its two halves are H(0) and H(1).)
        constexpr unsigned hbytes = (unsigned)(G::N / 2) * SB;
        cx h0[8], h1[8], hn[16];
#pragma unroll
        for (int r = 0; r < 8; r++) {
          h0[r] = ua[r];
          h1[r] = ua[8 + r];
        }
        for (int f = 0; f < K; f++) {
#pragma unroll
          for (int r = 0; r < 8; r++) {
            ub[r] = h0[r];
            ub[8 + r] = h1[r];
          }
          // H(f+2) is fetched from inside frame f's first pass, one load per radix-4 group
          frame_compute<C, true, true>(ub, c, f, &hn, rsrc, voff, f + 1 < K ? (unsigned)(f + 2) * hbytes : kNowhere);
#pragma unroll
          for (int r = 0; r < 8; r++) {
            h0[r] = h1[r];
            h1[r] = hn[r];
          }
        }
        epoch_close<C>(c, p, epoch_base);
        return;
      }
    }
    if constexpr ((C::OPT & kPair) != 0 && C::ABL == 0 && C::NBUF == 2) {
      // Frame pairs, two pairs per iteration in ping-pong: (ua, ub) and (uc, ud).
      cx uc[16], ud[16];
      load_frame<R3, NT, SC>(ub, rsrc, voff, K > 1 ? fbytes : kNowhere);
      int f = 0;
      for (; f + 3 < K; f += 4) {
        load_frame<R3, NT, SC>(uc, rsrc, voff, (unsigned)(f + 2) * fbytes);
        load_frame<R3, NT, SC>(ud, rsrc, voff, (unsigned)(f + 3) * fbytes);
        frame_pair_compute<C>(ua, ub, c);
        load_frame<R3, NT, SC>(ua, rsrc, voff, f + 4 < K ? (unsigned)(f + 4) * fbytes : kNowhere);
        load_frame<R3, NT, SC>(ub, rsrc, voff, f + 5 < K ? (unsigned)(f + 5) * fbytes : kNowhere);
        frame_pair_compute<C>(uc, ud, c);
      }
      const int rem = K - f;  // 0..3 frames left, the first two of them already in (ua, ub)
      if (rem >= 2) {
        load_frame<R3, NT, SC>(uc, rsrc, voff, rem == 3 ? (unsigned)(f + 2) * fbytes : kNowhere);
        frame_pair_compute<C>(ua, ub, c);
        if (rem == 3) {
          group_sync<C>();
          frame_compute<C>(uc, c, 0);
        }
      } else if (rem == 1) {
        group_sync<C>();
        frame_compute<C>(ua, c, 0);
      }
      epoch_close<C>(c, p, epoch_base);
      return;
    }
    if constexpr ((C::OPT & kSpread) != 0 && (C::OPT & kMulti) == 0 && C::ABL == 0 && C::PREFETCH) {
      // One epoch group per workgroup; frame f+1's loads are issued from inside frame f's butterflies.
      int f = 0;
      for (; f + 1 < K; f += 2) {
        frame_compute<C, true>(ua, c, f, &ub, rsrc, voff, (unsigned)(f + 1) * fbytes);
        frame_compute<C, true>(ub, c, f + 1, &ua, rsrc, voff, f + 2 < K ? (unsigned)(f + 2) * fbytes : kNowhere);
      }
      if (f < K) frame_compute<C>(ua, c, f);
      epoch_close<C>(c, p, epoch_base);
      return;
    }
    if constexpr ((C::OPT & kSpread) != 0 && (C::OPT & kMulti) != 0 && C::ABL == 0 && C::PREFETCH) {
      // This workgroup owns p.groups_per_wg consecutive epoch groups and treats their frames as
      // one stream: twiddles are loaded once, and the first frame of the next epoch is already in
      // flight while the last frame of this one is computed and closed (the per-workgroup prologue
      // and the exposed first load cost ~6 % at one epoch per workgroup).  Two register sets in
      // ping-pong; frame f+1's loads are issued from inside frame f's butterflies.
      // Workgroups are dispatched in blockIdx order; the last ones take a single epoch group, so the
      // machine drains in steps of one epoch instead of one 4-epoch workgroup (measured with
      // s_memrealtime stamps: the last 1024 workgroups used to finish spread over 200 us of a
      // 1.4 ms kernel).
      const StreamSpan sp = stream_span<R3>(p);
      const int epw = sp.epw, n_local = sp.n_local;
      const long long g0 = sp.g0;
      const __amdgpu_buffer_rsrc_t rs = group_rsrc<R3, (int)SB>(p, g0, epw);
      const unsigned gbytes = (unsigned)(G::GROUPS * (unsigned)p.epoch_stride) * SB;
      load_frame<R3, NT, SC>(ua, rs, voff, 0u, C::FULL ? G::N : c.L);
      int j = 0, f = 0;
#define CRN_STREAM_STEP(CUR, NXT)                                                                   \
      {                                                                                             \
        const bool last = f + 1 == K;                                                               \
        const int j_n = last ? j + 1 : j;                                                           \
        const int f_n = last ? 0 : f + 1;                                                           \
        const unsigned soff_n = j_n < n_local ? (unsigned)j_n * gbytes + (unsigned)f_n * fbytes : kNowhere; \
        frame_compute<C, true>(CUR, c, f, &NXT, rs, voff, soff_n);                                   \
        if (last) {                                                                                 \
          epoch_close<C>(c, p, (g0 + j) * G::GROUPS);                                               \
        }                                                                                           \
        j = j_n;                                                                                    \
        f = f_n;                                                                                    \
      }
      while (true) {
        CRN_STREAM_STEP(ua, ub)
        if (j >= n_local) break;
        CRN_STREAM_STEP(ub, ua)
        if (j >= n_local) break;
      }
#undef CRN_STREAM_STEP
      return;
    }
    if constexpr (C::PREFETCH && C::ABL < 2) {
      // Two register sets in ping-pong: while frame f is computed from one set, frame f+1 lands in
      // the other.  Always 16 loads per step, so the compiler waits with a counted vmcnt; after the
      // last frame they point outside the window and fetch nothing.
      int f = 0;
      for (; f + 1 < K; f += 2) {
        load_frame<R3, NT, SC>(ub, rsrc, voff, (unsigned)(f + 1) * fbytes);
        frame_step<C>(ua, c, f, u0);
        load_frame<R3, NT, SC>(ua, rsrc, voff, f + 2 < K ? (unsigned)(f + 2) * fbytes : kNowhere);
        frame_step<C>(ub, c, f + 1, u0);
      }
      if (f < K) frame_step<C>(ua, c, f, u0);
    } else {
      for (int f = 0; f < K; f++) {
        frame_step<C>(ua, c, f, u0);
        if constexpr (C::ABL < 2)
          load_frame<R3, NT, SC>(ua, rsrc, voff, f + 1 < K ? (unsigned)(f + 1) * fbytes : kNowhere);
      }
    }
    epoch_close<C>(c, p, epoch_base);
  }
}

// ---------------------------------------------------------------------------------------------
// launch dispatch
// ---------------------------------------------------------------------------------------------
template <class C>
static hipError_t launch_cfg(const SenseParams &p, hipStream_t stream) {
  using G = Geo<C::R3>;
  const long long n_groups = (p.n_epochs + G::GROUPS - 1) / G::GROUPS;
  // Welch (hop = N/2) streams when the epochs are dense (see sense_kernel)
  const bool welch = C::WIN && p.frame_stride * 2 == G::N;
  const bool welch_stream = welch && p.epoch_stride == (long long)p.K * (G::N / 2);
  const bool multi = (C::OPT & kSpread) != 0 && (C::OPT & kMulti) != 0 && C::ABL == 0 && C::PREFETCH &&
                     (!welch || welch_stream);
  SenseParams q = p;
  unsigned grid;
  if (multi) {
    // n_big_wgs workgroups of groups_per_wg groups, then one workgroup per remaining group
    if (q.n_big_wgs * q.groups_per_wg > n_groups) q.n_big_wgs = n_groups / q.groups_per_wg;
    grid = (unsigned)(q.n_big_wgs + (n_groups - q.n_big_wgs * q.groups_per_wg));
  } else {
    q.n_big_wgs = 0;
    grid = (unsigned)n_groups;
  }
  const size_t lds = ((size_t)G::GROUPS * C::NBUF * G::GROUP_CPLX + 16 * C::R3) * sizeof(cx) + kCloseLdsBytes;
  if (grid == 0) return hipSuccess;
  auto kfn = sense_kernel<C>;
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kfn),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(kfn, dim3(grid), dim3(256), lds, stream, q);
  return hipGetLastError();
}

// The register form of the epoch close applies to band plans the host could cut into row entries
// (crn_api.cpp) when no per-bin spectrum is stored.
static bool reg_bands(const SenseParams &p) { return p.n_row_entries > 0 && p.spectrum == nullptr; }

// Default configuration of every size: all mode / window / short-frame combinations.
template <int R3, int NBUF, bool PREFETCH, bool NT, bool TW2LDS, int OCC, bool PK, int OPT = kSpread | kLdsBlk | kPrioValu | kMulti,
          int WHICH = 0 /* 0 all, 1 unwindowed kernels only, 2 windowed only */>
static hipError_t launch_default(const SenseParams &p, bool mag, bool win, hipStream_t stream) {
  const bool full = p.L == Geo<R3>::N;
  const bool regb = reg_bands(p);  // small band plan, no spectrum: band sums from registers
#define CRN_GO(MAGV, WINV, FULLV) return launch_cfg<Cfg<R3, NBUF, PREFETCH, NT, MAGV, WINV, TW2LDS, OCC, 0, FULLV, PK, OPT>>(p, stream)
#define CRN_GO_R(MAGV, WINV, FULLV) return launch_cfg<Cfg<R3, NBUF, PREFETCH, NT, MAGV, WINV, TW2LDS, OCC, 0, FULLV, PK, OPT | kRegBands>>(p, stream)
  if constexpr (WHICH != 1) {
    if (mag && win) { if (full) CRN_GO(true, true, true); else CRN_GO(true, true, false); }
    if (win) { if (full) CRN_GO(false, true, true); else CRN_GO(false, true, false); }
  }
  if constexpr (WHICH != 2) {
    if (regb) {
      if (mag) { if (full) CRN_GO_R(true, false, true); else CRN_GO_R(true, false, false); }
      if (full) CRN_GO_R(false, false, true);
      CRN_GO_R(false, false, false);
    }
    if (mag) { if (full) CRN_GO(true, false, true); else CRN_GO(true, false, false); }
    if (full) CRN_GO(false, false, true);
    CRN_GO(false, false, false);
  }
  return hipErrorInvalidValue;
#undef CRN_GO
#undef CRN_GO_R
}

// A/B variants: compiled for the headline shape only (N = 4096, energy mode, no window, L = N).
template <int R3, int NBUF, bool PREFETCH, bool NT, bool TW2LDS, int OCC, int ABL, bool PK, int OPT = 0>
static hipError_t launch_rn(const SenseParams &p, bool, bool, hipStream_t stream) {
  return launch_cfg<Cfg<R3, NBUF, PREFETCH, NT, false, false, TW2LDS, OCC, ABL, true, PK, OPT>>(p, stream);
}


}  // namespace crn
#endif
