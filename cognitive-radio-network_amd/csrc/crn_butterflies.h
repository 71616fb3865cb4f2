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
// the transforms below also compile for the host (their scalar form, PK = false): tests/harness/butterfly_unit.cpp checks the
// algebra against a double-precision DFT without a GPU
#define CRN_HD static __host__ __device__ __forceinline__

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
#define CRN_C1 0.92387953251128674f  // cos(pi/8)
#define CRN_S1 0.38268343236508977f  // sin(pi/8)
#define CRN_H_ 0.70710678118654752f  // sqrt(1/2)

template <bool PK>
struct M {
  CRN_HD cx add(cx a, cx b) {
    if constexpr (PK) { cx d; asm("v_pk_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
    else return cx{a.x + b.x, a.y + b.y};
  }
  CRN_HD cx sub(cx a, cx b) {
    if constexpr (PK) { cx d; asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b)); return d; }
    else return cx{a.x - b.x, a.y - b.y};
  }
  // a + (-j) b = (a.x + b.y, a.y - b.x)
  CRN_HD cx add_mj(cx a, cx b) {
    if constexpr (PK) { cx d; asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b)); return d; }
    else return cx{a.x + b.y, a.y - b.x};
  }
  // a - (-j) b = (a.x - b.y, a.y + b.x)
  CRN_HD cx sub_mj(cx a, cx b) {
    if constexpr (PK) { cx d; asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(d) : "v"(a), "v"(b)); return d; }
    else return cx{a.x - b.y, a.y + b.x};
  }
  // a * w, w in VGPRs (per-lane twiddle)
  CRN_HD cx mul(cx a, cx w) {
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
  CRN_HD cx mul_conj(cx a, cx w) {
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
  CRN_HD cx fma_w(cx wp, cx t, cx x) {
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
  // The twiddles W16^2 = sqrt(1/2) (1 - j), W16^6 = -sqrt(1/2) (1 + j) (and W8^1, W8^3) are not multiplied out: (1 -+ j) a is one
  // packed add of a with itself rotated (add_mj / sub_mj below), and the factor h = sqrt(1/2) rides in the FMA that consumes the
  // product — x +- h u instead of x +- (u * w): one packed add + the FMA where a complex multiply (two packed instructions, four
  // real multiplies) + an add stood.  4 packed instructions fewer per 16-point transform, 12 of the ~334 of a 4096-point frame.
  //   fma_h(u, x) = x + h u      fms_h(u, x) = x - h u      fma_h_mj(u, x) = x + (-j) h u      fms_h_mj(u, x) = x - (-j) h u
  // (h comes as an SGPR pair (h, h); neg_* on source 1 flips the sign of h for one or both halves)
  CRN_HD cx fma_h(cx u, cx x) {
    if constexpr (PK) { cx d; const cx h = {CRN_H_, CRN_H_}; asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(u), "s"(h), "v"(x)); return d; }
    else return cx{fmaf(CRN_H_, u.x, x.x), fmaf(CRN_H_, u.y, x.y)};
  }
  CRN_HD cx fms_h(cx u, cx x) {
    if constexpr (PK) { cx d; const cx h = {CRN_H_, CRN_H_}; asm("v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[0,1,0] neg_hi:[0,1,0]" : "=v"(d) : "v"(u), "s"(h), "v"(x)); return d; }
    else return cx{fmaf(-CRN_H_, u.x, x.x), fmaf(-CRN_H_, u.y, x.y)};
  }
  // x + (-j) h u = (x.x + h u.y, x.y - h u.x)
  CRN_HD cx fma_h_mj(cx u, cx x) {
    if constexpr (PK) {
      cx d; const cx h = {CRN_H_, CRN_H_};
      asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[0,1,1] neg_hi:[0,1,0]" : "=v"(d) : "v"(u), "s"(h), "v"(x));
      return d;
    } else return cx{fmaf(CRN_H_, u.y, x.x), fmaf(-CRN_H_, u.x, x.y)};
  }
  // x - (-j) h u = (x.x - h u.y, x.y + h u.x)
  CRN_HD cx fms_h_mj(cx u, cx x) {
    if constexpr (PK) {
      cx d; const cx h = {CRN_H_, CRN_H_};
      asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]" : "=v"(d) : "v"(u), "s"(h), "v"(x));
      return d;
    } else return cx{fmaf(-CRN_H_, u.y, x.x), fmaf(CRN_H_, u.x, x.y)};
  }
  // a * w, w a wave-uniform constant held in an SGPR pair
  CRN_HD cx mul_c(cx a, cx w) {
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
CRN_HD void static_for_(F &f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for_<I + 1, N>(f);
  }
}
template <int N, class F>
CRN_HD void static_for(F &&f) {
  static_for_<0, N>(f);
}


// 4-point forward DFT in place.  B2MJ: input a2 still lacks a factor -j (a folded W16^4 / W8^2).
template <bool PK, bool B2MJ = false>
CRN_HD void dft4(cx &a0, cx &a1, cx &a2, cx &a3) {
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
  __host__ __device__ __forceinline__ void operator()(int) const {}
};

// Twiddles W16^{r0 a0} and level B of the 16-point transform (shared by the plain and the windowed level A).
template <bool PK, class Hook = NoHook>
CRN_HD void dft16_level_b(cx (&y)[16], cx (&out)[16], const Hook &hook = Hook()) {
  using m = M<PK>;
  // W16^{r0 a0} on element (a0, r0) = y[r0 + 4 a0], then for each a0 a DFT4 over r0; X[a0 + 4 a1] = y[a1 + 4 a0].
  // W16^4 = -j is folded into the butterfly; W16^2 and W16^6 are (1 - j) / -(1 + j) times h = sqrt(1/2): one rotated add each, h
  // applied by the FMAs that consume them (see M::fma_h).  W16^1, W16^3, W16^9 are multiplied out.
  const cx w1 = {CRN_C1, -CRN_S1}, w3 = {CRN_S1, -CRN_C1}, w9 = {-CRN_C1, CRN_S1};
  dft4<PK>(y[0], y[1], y[2], y[3]);                      // a0 = 0: no twiddles
  hook(4);
  {                                                      // a0 = 1: W16^1, W16^2 = h (1 - j), W16^3
    const cx b0 = y[4], b1 = m::mul_c(y[5], w1), b3 = m::mul_c(y[7], w3);
    const cx p = m::add_mj(y[6], y[6]);                  // (1 - j) y
    const cx s02 = m::fma_h(p, b0), d02 = m::fms_h(p, b0);
    const cx s13 = m::add(b1, b3), d13 = m::sub(b1, b3);
    y[4] = m::add(s02, s13);
    y[6] = m::sub(s02, s13);
    y[5] = m::add_mj(d02, d13);
    y[7] = m::sub_mj(d02, d13);
  }
  hook(5);
  {                                                      // a0 = 2: W16^2 = h (1 - j), W16^4 = -j, W16^6 = -h (1 + j)
    const cx p = m::add_mj(y[9], y[9]), q = m::sub_mj(y[11], y[11]);   // (1 - j) y1, (1 + j) y3
    const cx s02 = m::add_mj(y[8], y[10]), d02 = m::sub_mj(y[8], y[10]);
    const cx u = m::sub(p, q), v = m::add(p, q);         // s13 = h u, d13 = h v
    y[8] = m::fma_h(u, s02);
    y[10] = m::fms_h(u, s02);
    y[9] = m::fma_h_mj(v, d02);
    y[11] = m::fms_h_mj(v, d02);
  }
  hook(6);
  {                                                      // a0 = 3: W16^3, W16^6 = -h (1 + j), W16^9
    const cx b0 = y[12], b1 = m::mul_c(y[13], w3), b3 = m::mul_c(y[15], w9);
    const cx q = m::sub_mj(y[14], y[14]);                // (1 + j) y
    const cx s02 = m::fms_h(q, b0), d02 = m::fma_h(q, b0);
    const cx s13 = m::add(b1, b3), d13 = m::sub(b1, b3);
    y[12] = m::add(s02, s13);
    y[14] = m::sub(s02, s13);
    y[13] = m::add_mj(d02, d13);
    y[15] = m::sub_mj(d02, d13);
  }
  hook(7);
#pragma unroll
  for (int a0 = 0; a0 < 4; a0++)
#pragma unroll
    for (int a1 = 0; a1 < 4; a1++) out[a0 + 4 * a1] = y[a1 + 4 * a0];
}

// 16-point forward DFT, natural order in and out, as 4 x 4.  `hook(k)`, k = 0..7, runs after the
// k-th radix-4 group: the caller uses it to drop one prefetch load into the butterfly stream.
template <bool PK, class Hook = NoHook>
CRN_HD void dft16(const cx (&in)[16], cx (&out)[16], const Hook &hook = Hook()) {
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
CRN_HD void dft16_hann(const cx (&in)[16], cx (&out)[16], const cx (&wp)[4], const Hook &hook = Hook()) {
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

// The same at every size.  After pass 3 thread (a, g) holds bin a + 16 (g J + j) + 256 d in accumulator register j R3 + d
// (J = 16 / R3, N = 256 R3): which of the 16 registers can hold a bin of the reference's channel plan scaled to N points — bit
// j R3 + d.  N = 4096: the rows above (J = 1: register = row), 7 of 16; N = 512: 7; N = 1024: 12; N = 2048: 11.  Band tables inside
// the mask, with no per-bin spectrum asked for, run kernels whose pass 3 forms and accumulates only those registers.
constexpr unsigned ref_acc_mask(int R3) {
  const int seg[5][2] = {{0, 16}, {496, 511}, {55, 85}, {189, 222}, {300, 310}};   // CE_Predictive_Node.cpp:173-191, of 512 bins
  const int J = 16 / R3, S = R3 / 2;                                               // N / 512
  unsigned mask = 0;
  for (int s = 0; s < 5; s++)
    for (int k = seg[s][0] * S; k < seg[s][1] * S; k++) mask |= 1u << ((((k & 255) >> 4) % J) * R3 + (k >> 8));
  return mask;
}
static_assert(ref_acc_mask(16) == kRefPlanRows, "N = 4096: register = 256-bin row");

// DFT16 whose last level only forms the outputs named in MASK (bit d = X[d] needed).
template <bool PK, unsigned MASK>
CRN_HD void dft16_pruned(const cx (&in)[16], cx (&out)[16]) {
  using m = M<PK>;
  cx y[16];
#pragma unroll
  for (int i = 0; i < 16; i++) y[i] = in[i];
#pragma unroll
  for (int r0 = 0; r0 < 4; r0++) dft4<PK>(y[r0], y[r0 + 4], y[r0 + 8], y[r0 + 12]);
  // level B as in dft16_level_b (same twiddle forms, same roundings for the outputs that are formed), output d = a0 + 4 a1 only when
  // MASK names it: out[a0] / out[a0 + 8] need (s02, s13), out[a0 + 4] / out[a0 + 12] need (d02, d13)
  const cx w1 = {CRN_C1, -CRN_S1}, w3 = {CRN_S1, -CRN_C1}, w9 = {-CRN_C1, CRN_S1};
  static_for<4>([&](auto ac) {
    constexpr int a0 = decltype(ac)::value;
    constexpr bool n0 = (MASK >> (a0 + 0)) & 1, n1 = (MASK >> (a0 + 4)) & 1, n2 = (MASK >> (a0 + 8)) & 1, n3 = (MASK >> (a0 + 12)) & 1;
    constexpr bool S = n0 || n2, D = n1 || n3;
    const cx b0 = y[4 * a0];
    if constexpr (a0 == 2) {   // W16^2 = h (1 - j), W16^4 = -j, W16^6 = -h (1 + j): s13 = h u, d13 = h v
      const cx p = m::add_mj(y[9], y[9]), q = m::sub_mj(y[11], y[11]);
      if constexpr (S) {
        const cx s02 = m::add_mj(b0, y[10]), u = m::sub(p, q);
        if constexpr (n0) out[a0 + 0] = m::fma_h(u, s02);
        if constexpr (n2) out[a0 + 8] = m::fms_h(u, s02);
      }
      if constexpr (D) {
        const cx d02 = m::sub_mj(b0, y[10]), v = m::add(p, q);
        if constexpr (n1) out[a0 + 4] = m::fma_h_mj(v, d02);
        if constexpr (n3) out[a0 + 12] = m::fms_h_mj(v, d02);
      }
    } else {
      cx b1 = y[4 * a0 + 1], b3 = y[4 * a0 + 3];
      if constexpr (a0 == 1) { b1 = m::mul_c(b1, w1); b3 = m::mul_c(b3, w3); }
      if constexpr (a0 == 3) { b1 = m::mul_c(b1, w3); b3 = m::mul_c(b3, w9); }
      const cx r = a0 == 1 ? m::add_mj(y[4 * a0 + 2], y[4 * a0 + 2]) : a0 == 3 ? m::sub_mj(y[4 * a0 + 2], y[4 * a0 + 2]) : y[4 * a0 + 2];
      if constexpr (S) {
        const cx s02 = a0 == 0 ? m::add(b0, r) : a0 == 1 ? m::fma_h(r, b0) : m::fms_h(r, b0);
        const cx s13 = m::add(b1, b3);
        if constexpr (n0) out[a0 + 0] = m::add(s02, s13);
        if constexpr (n2) out[a0 + 8] = m::sub(s02, s13);
      }
      if constexpr (D) {
        const cx d02 = a0 == 0 ? m::sub(b0, r) : a0 == 1 ? m::fms_h(r, b0) : m::fma_h(r, b0);
        const cx d13 = m::sub(b1, b3);
        if constexpr (n1) out[a0 + 4] = m::add_mj(d02, d13);
        if constexpr (n3) out[a0 + 12] = m::sub_mj(d02, d13);
      }
    }
  });
}

// 4-point transform that forms only the outputs named in MSK (bit k = a_k needed); the ones it forms are dft4's, bit for bit.
template <bool PK, unsigned MSK, bool B2MJ = false>
CRN_HD void dft4_pruned(cx &a0, cx &a1, cx &a2, cx &a3) {
  using m = M<PK>;
  constexpr bool S = (MSK & 5u) != 0, D = (MSK & 10u) != 0;   // outputs 0 / 2 need the sums, 1 / 3 the differences
  cx s02 = a0, d02 = a0, s13 = a1, d13 = a1;
  if constexpr (S) { s02 = B2MJ ? m::add_mj(a0, a2) : m::add(a0, a2); s13 = m::add(a1, a3); }
  if constexpr (D) { d02 = B2MJ ? m::sub_mj(a0, a2) : m::sub(a0, a2); d13 = m::sub(a1, a3); }
  if constexpr ((MSK & 1u) != 0) a0 = m::add(s02, s13);
  if constexpr ((MSK & 4u) != 0) a2 = m::sub(s02, s13);
  if constexpr ((MSK & 2u) != 0) a1 = m::add_mj(d02, d13);
  if constexpr ((MSK & 8u) != 0) a3 = m::sub_mj(d02, d13);
}

// 8-point transform that forms only the outputs named in M8 (bit d); same operations as dft8 for those.
template <bool PK, unsigned M8>
CRN_HD void dft8_pruned(const cx (&in)[8], cx (&out)[8]) {
  using m = M<PK>;
  cx y[8];
#pragma unroll
  for (int i = 0; i < 8; i++) y[i] = in[i];
  constexpr unsigned need = (M8 | (M8 >> 4)) & 15u;   // E[a0] and O[a0] are needed when output a0 or a0 + 4 is
  dft4_pruned<PK, need>(y[0], y[2], y[4], y[6]);
  dft4_pruned<PK, need>(y[1], y[3], y[5], y[7]);
  if constexpr ((M8 & 0x01u) != 0) out[0] = m::add(y[0], y[1]);
  if constexpr ((M8 & 0x10u) != 0) out[4] = m::sub(y[0], y[1]);
  if constexpr ((M8 & 0x22u) != 0) {
    const cx p = m::add_mj(y[3], y[3]);
    if constexpr ((M8 & 0x02u) != 0) out[1] = m::fma_h(p, y[2]);
    if constexpr ((M8 & 0x20u) != 0) out[5] = m::fms_h(p, y[2]);
  }
  if constexpr ((M8 & 0x04u) != 0) out[2] = m::add_mj(y[4], y[5]);
  if constexpr ((M8 & 0x40u) != 0) out[6] = m::sub_mj(y[4], y[5]);
  if constexpr ((M8 & 0x88u) != 0) {
    const cx q = m::sub_mj(y[7], y[7]);
    if constexpr ((M8 & 0x08u) != 0) out[3] = m::fms_h(q, y[6]);
    if constexpr ((M8 & 0x80u) != 0) out[7] = m::fma_h(q, y[6]);
  }
}

// 8-point forward DFT as 2 x 4.
template <bool PK>
CRN_HD void dft8(const cx (&in)[8], cx (&out)[8]) {
  using m = M<PK>;
  cx y[8];
#pragma unroll
  for (int i = 0; i < 8; i++) y[i] = in[i];
  dft4<PK>(y[0], y[2], y[4], y[6]);  // r = r0 + 2 r1: DFT4 over r1 -> a0 at y[r0 + 2 a0]
  dft4<PK>(y[1], y[3], y[5], y[7]);
  // W8^1 = h (1 - j) on (r0 = 1, a0 = 1), W8^3 = -h (1 + j) on (r0 = 1, a0 = 3): a rotated add, h applied by the FMA (M::fma_h);
  // W8^2 = -j on y[5] folded into its butterfly
  const cx p = m::add_mj(y[3], y[3]), q = m::sub_mj(y[7], y[7]);
  out[0] = m::add(y[0], y[1]);
  out[4] = m::sub(y[0], y[1]);
  out[1] = m::fma_h(p, y[2]);
  out[5] = m::fms_h(p, y[2]);
  out[2] = m::add_mj(y[4], y[5]);
  out[6] = m::sub_mj(y[4], y[5]);
  out[3] = m::fms_h(q, y[6]);
  out[7] = m::fma_h(q, y[6]);
}

}  // namespace crn
#endif
