// crn_butterflies.h — complex arithmetic on packed-f32 register pairs and the 4 / 8 / 16-point transforms the three register passes of the sensing
// kernel are made of (csrc/crn_frame.h).  Internal linkage throughout.
#ifndef CRN_BUTTERFLIES_H
#define CRN_BUTTERFLIES_H
#include <hip/hip_runtime.h>
#include <type_traits>
#include <stdint.h>

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

}  // namespace crn
#endif
