// butterfly_unit.cpp — TEST INFRASTRUCTURE.  The 4 / 8 / 16-point transforms of csrc/crn_butterflies.h (their scalar form, PK = false,
// compiled for the HOST) against a double-precision DFT: the algebra of the hand-placed twiddles — W16^4 = -j folded into a butterfly,
// W16^2 / W16^6 / W8^1 / W8^3 as a rotated add whose sqrt(1/2) rides in the consuming FMA, the Hann window folded into the first
// butterflies, the pruned last level — without a GPU.  (The packed-f32 inline-asm forms of the same formulas are what the GPU parity
// tests run.)
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include <complex>
#include <random>

#include "../../cognitive-radio-network_amd/csrc/crn_butterflies.h"

using crn::cx;
typedef std::complex<double> zd;

static double worst = 0.0;
static void check(const char *what, const cx *got, const zd *want, int n, unsigned mask = 0xFFFFFFFFu) {
  double scale = 0;
  for (int k = 0; k < n; k++) scale = std::max(scale, std::abs(want[k]));
  if (scale == 0.0) scale = 1.0;   // an all-zero input: the outputs must be exactly zero
  for (int k = 0; k < n; k++) {
    if (!((mask >> k) & 1)) continue;
    const double e = std::abs(zd(got[k].x, got[k].y) - want[k]) / scale;
    worst = std::max(worst, e);
    if (!(e < 2e-6)) {   // fp32: a handful of roundings on values of magnitude <= scale
      fprintf(stderr, "butterfly_unit: %s: output %d off by %.3g of the largest output\n", what, k, e);
      exit(1);
    }
  }
}
static void dft(const zd *x, zd *X, int n) {
  for (int k = 0; k < n; k++) {
    zd s = 0;
    for (int i = 0; i < n; i++) s += x[i] * std::polar(1.0, -2.0 * M_PI * (double)((long)k * i % n) / n);
    X[k] = s;
  }
}

int main() {
  std::mt19937 rng(12345);
  std::normal_distribution<float> g(0.f, 1.f);
  for (int trial = 0; trial < 2000; trial++) {
    cx in[16], out[16];
    zd x[16], X[16];
    for (int i = 0; i < 16; i++) {
      in[i] = cx{g(rng), g(rng)};
      if (trial % 7 == 0) in[i] = cx{i == trial % 16 ? 1.f : 0.f, 0.f};   // impulses: every twiddle path on its own
      x[i] = zd(in[i].x, in[i].y);
    }
    // 4 points, plain and with the folded -j on input 2
    {
      cx a[4] = {in[0], in[1], in[2], in[3]};
      crn::dft4<false>(a[0], a[1], a[2], a[3]);
      dft(x, X, 4);
      check("dft4", a, X, 4);
      cx b[4] = {in[0], in[1], in[2], in[3]};
      crn::dft4<false, true>(b[0], b[1], b[2], b[3]);
      zd y[4] = {x[0], x[1], x[2] * zd(0, -1), x[3]};
      dft(y, X, 4);
      check("dft4 with -j folded", b, X, 4);
    }
    {
      cx i8[8], o8[8];
      for (int i = 0; i < 8; i++) i8[i] = in[i];
      crn::dft8<false>(i8, o8);
      dft(x, X, 8);
      check("dft8", o8, X, 8);
    }
    {
      // the pruned 4- and 8-point forms: whatever subset of outputs they are asked for is the full transform's, bit for bit
      cx f4[4] = {in[0], in[1], in[2], in[3]};
      crn::dft4<false>(f4[0], f4[1], f4[2], f4[3]);
      cx f8[8], i8[8];
      for (int i = 0; i < 8; i++) i8[i] = in[i];
      crn::dft8<false>(i8, f8);
      crn::static_for<15>([&](auto mc) {
        constexpr unsigned M4 = decltype(mc)::value + 1;
        cx p4[4] = {in[0], in[1], in[2], in[3]};
        crn::dft4_pruned<false, M4>(p4[0], p4[1], p4[2], p4[3]);
        for (int k = 0; k < 4; k++)
          if (((M4 >> k) & 1) && (p4[k].x != f4[k].x || p4[k].y != f4[k].y)) { fprintf(stderr, "butterfly_unit: dft4_pruned<%u> output %d\n", M4, k); exit(1); }
      });
      crn::static_for<255>([&](auto mc) {
        constexpr unsigned M8 = decltype(mc)::value + 1;
        cx p8[8];
        for (int i = 0; i < 8; i++) p8[i] = cx{0.f, 0.f};
        crn::dft8_pruned<false, M8>(i8, p8);
        for (int k = 0; k < 8; k++)
          if (((M8 >> k) & 1) && (p8[k].x != f8[k].x || p8[k].y != f8[k].y)) { fprintf(stderr, "butterfly_unit: dft8_pruned<%u> output %d\n", M8, k); exit(1); }
      });
    }
    crn::dft16<false>(in, out);
    dft(x, X, 16);
    check("dft16", out, X, 16);
    {
      constexpr unsigned MASK = crn::kRefPlanRows;
      cx po[16];
      for (int i = 0; i < 16; i++) po[i] = cx{0.f, 0.f};
      crn::dft16_pruned<false, MASK>(in, po);
      check("dft16_pruned", po, X, 16, MASK);
      for (int d = 0; d < 16; d++)      // the outputs it forms are the full transform's, bit for bit (same operations, same order)
        if (((MASK >> d) & 1) && (po[d].x != out[d].x || po[d].y != out[d].y)) {
          fprintf(stderr, "butterfly_unit: dft16_pruned output %d differs from dft16's\n", d);
          return 1;
        }
    }
    {
      // periodic Hann folded into level A: rows r and r + 8 share one weight w (w[n + N/2] = 1 - w[n])
      cx wp[4];
      float w[8];
      zd xw[16];
      for (int r = 0; r < 8; r++) w[r] = 0.5f - 0.5f * cosf(2.f * (float)M_PI * ((float)r + 0.37f) / 16.f);
      for (int p = 0; p < 4; p++) wp[p] = cx{w[2 * p], w[2 * p + 1]};
      for (int r = 0; r < 8; r++) {
        xw[r] = x[r] * (double)w[r];
        xw[r + 8] = x[r + 8] * (1.0 - (double)w[r]);
      }
      cx ho[16];
      crn::dft16_hann<false>(in, ho, wp);
      dft(xw, X, 16);
      check("dft16_hann", ho, X, 16);
    }
  }
  // which accumulator registers the reference channel plan reaches, per size (csrc/crn_butterflies.h: ref_acc_mask): restated here from
  // the bin ranges of CE_Predictive_Node.cpp:173-191 by walking every thread's bins
  for (int R3 : {2, 4, 8, 16}) {
    const int N = 256 * R3, J = 16 / R3, S = N / 512;
    const int seg[5][2] = {{0, 16}, {496, 511}, {55, 85}, {189, 222}, {300, 310}};
    unsigned mask = 0;
    for (int a = 0; a < 16; a++)
      for (int g = 0; g < R3; g++)
        for (int j = 0; j < J; j++)
          for (int d = 0; d < R3; d++) {
            const int k = a + 16 * (g * J + j) + 256 * d;   // the bin thread (a, g) holds in register j R3 + d
            for (int sgi = 0; sgi < 5; sgi++)
              if (k >= seg[sgi][0] * S && k < seg[sgi][1] * S) mask |= 1u << (j * R3 + d);
          }
    if (mask != crn::ref_acc_mask(R3)) { fprintf(stderr, "butterfly_unit: ref_acc_mask(%d) = %#x, walking the bins gives %#x\n", R3, crn::ref_acc_mask(R3), mask); return 1; }
    printf("butterfly_unit: N = %4d: the reference channel plan reaches %2d of 16 accumulator registers (mask %#06x)\n", N, __builtin_popcount(mask), mask);
  }
  printf("butterfly_unit: ok (worst error %.3g of the largest output)\n", worst);
  return 0;
}
