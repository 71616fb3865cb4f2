/*
 * crn_oracle.c — CPU restatement of the reference sensing path.  TEST INFRASTRUCTURE ONLY.
 *
 *   *** PARITY UNPINNED ***
 *   The reference (0xastro/Cognitive-Radio-Network) ships no test, golden vector or fixture for
 *   this path, and it cannot be built in this image: CE_Predictive_Node.hpp:4 includes
 *   <liquid/liquid.h> and extensible_cognitive_radio.hpp:9-11 includes UHD headers; liquid-dsp,
 *   UHD and libconfig are neither vendored nor installed, and writing stand-ins for them is not
 *   allowed.  So this restatement is never checked against reference output.  What it IS checked
 *   against (tests/test_golden.py, tests/test_oracle.py): fixtures written by an independent float64
 *   numpy statement of the same mathematics that shares no code with this file (tests/ref_f64.py ->
 *   tests/golden/), SURVEY.md Appendix C's probe values, the DFT definition in float64, and
 *   hand-derived known answers.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this file's
 * shared object.  The product (libcrnsense) never links or calls it.
 *
 * What is restated, and from where (paths relative to the reference root):
 *   - crn_oracle_ref_epoch(): cognitive_engines/CE_Predictive_Node/CE_Predictive_Node.cpp
 *     lines 146-289 and CE_Predictive_Node.hpp lines 20-22, 30-33, 55-57, for ONE decision epoch.
 *   - crn_oracle_fft_radix2(): the third-party FFT behind fft_execute (.cpp:150).  It lives in
 *     liquid-dsp, pinned by HardwareSetup/Install_liquid-dsp.sh:36 to git a4d7c80d3, absent from
 *     the reference tree.  Restated from the published liquid-dsp source of that era,
 *     src/fft/src/fft_radix2.c (power-of-two plan: bit-reversed copy, then log2(N) in-place
 *     decimation-in-time passes over a twiddle table twiddle[i] = cexpf(-j*2*pi*i/N) whose angle
 *     is formed in double and narrowed to float before cexpf).  Unverifiable here; if liquid was
 *     configured against FFTW3 (the ECR's CE_fftw_mutex comment,
 *     include/extensible_cognitive_radio.hpp:880-884, suggests CORNET did) fft_execute is FFTW's
 *     and its rounding is different again.  The mathematical contract either way is the
 *     unnormalised forward DFT; parity tolerances are stated against float64.
 *   - crn_oracle_run(): the same pipeline generalised exactly as include/crn_sense.h documents
 *     (N, K, hop, window, band table, ENERGY mode, threshold decision).  With crn_cfg_reference
 *     parameters it must agree bit-for-bit with crn_oracle_ref_epoch (tests check this).
 *
 * Build: gcc -O2 -ffp-contract=off (no FMA contraction: the reference is built -O2/-g for
 * generic x86-64, makefile:1, where `a*b + c` is two roundings).
 */
#include <complex.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/crn_sense.h"

#define ORACLE_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------------------------
 * Reference constants (independent restatement; the product has its own copy in crn_cfg.cpp —
 * tests compare the two).
 * ---------------------------------------------------------------------------------------- */
enum { REF_N = 512, REF_K = 10 }; /* fft_length, fft_averaging: CE_Predictive_Node.hpp:31-32 */

/* CE_Predictive_Node.cpp:78-102 — WeightIH[i][j], i = 0 bias, 1 NF, 2 CH1, 3 CH2, 4 CH3. */
static const double REF_W_IH[5][6] = {
    {0, -0.188208, -0.170684, -0.024726, 0.001448, 0.015983},
    {0, -0.106634, -0.415470, 0.309261, 0.159974, 0.212781},
    {0, 0.005650, 0.741944, 0.006133, -0.620100, 0.669892},
    {0, -0.057578, 0.621154, -0.048268, -0.249186, 0.734475},
    {0, 0.092680, 0.809336, -0.010821, -0.546496, 0.609384},
};
/* CE_Predictive_Node.cpp:103-120 — WeightHO[j][k], j = 0 bias, 1..5 hidden. */
static const double REF_W_HO[6][4] = {
    {0, -7.033320, 2.726400, -2.590206},   {0, 10.857465, -18.452471, 15.609466},
    {0, -6.848443, 2.053071, -2.929559},   {0, 17.053079, -13.375309, -15.703407},
    {0, 0.087664, -0.269499, 0.407028},    {0, -6.552455, 2.655529, -2.552555},
};
/* CE_Predictive_Node.hpp:55-57 */
static const double REF_CHANNEL1 = 833e6, REF_CHANNEL2 = 835e6;

ORACLE_API void crn_oracle_ref_weights(double *w_ih /*5x6*/, double *w_ho /*6x4*/) {
  memcpy(w_ih, REF_W_IH, sizeof(REF_W_IH));
  memcpy(w_ho, REF_W_HO, sizeof(REF_W_HO));
}

/* ------------------------------------------------------------------------------------------
 * FFTs
 * ---------------------------------------------------------------------------------------- */
static unsigned bit_reverse(unsigned v, unsigned bits) {
  unsigned r = 0;
  for (unsigned b = 0; b < bits; b++) {
    r = (r << 1) | (v & 1u);
    v >>= 1;
  }
  return r;
}

static unsigned ilog2(unsigned n) {
  unsigned m = 0;
  while ((1u << m) < n) m++;
  return m;
}

typedef struct {
  unsigned n, m;
  unsigned *rev;
  float complex *tw;
} plan_f32;

/* fft_create_plan(n, x, y, LIQUID_FFT_FORWARD, 0) for n = 2^m — CE_Predictive_Node.cpp:42-45;
 * liquid-dsp fft_radix2.c plan creation (see header). */
static int plan_f32_init(plan_f32 *p, unsigned n) {
  p->n = n;
  p->m = ilog2(n);
  if ((1u << p->m) != n) return -1;
  p->rev = (unsigned *)malloc(n * sizeof(unsigned));
  p->tw = (float complex *)malloc(n * sizeof(float complex));
  if (!p->rev || !p->tw) return -1;
  for (unsigned i = 0; i < n; i++) p->rev[i] = bit_reverse(i, p->m);
  for (unsigned i = 0; i < n; i++) {
    /* angle in double, narrowed to float, then the float cexpf */
    double ang = -2.0 * M_PI * (double)i / (double)n;
    const float complex arg = CMPLXF(0.0f, (float)ang);
    p->tw[i] = cexpf(arg);
  }
  return 0;
}

static void plan_f32_free(plan_f32 *p) {
  free(p->rev);
  free(p->tw);
  p->rev = NULL;
  p->tw = NULL;
}

/* fft_execute — CE_Predictive_Node.cpp:150; liquid-dsp fft_radix2.c execute: y = bitrev(x);
 * for each of the m passes the butterfly span doubles and the twiddle stride halves. */
static void plan_f32_exec(const plan_f32 *p, const float complex *x, float complex *y) {
  const unsigned n = p->n;
  for (unsigned i = 0; i < n; i++) y[i] = x[p->rev[i]];
  unsigned half = 1, span = 2, stride = n >> 1;
  for (unsigned pass = 0; pass < p->m; pass++) {
    unsigned ti = 0;
    for (unsigned j = 0; j < half; j++) {
      const float complex t = p->tw[ti];
      ti = (ti + stride) % n;
      for (unsigned k = j; k < n; k += span) {
        const float complex yp = y[k + half] * t;
        y[k + half] = y[k] - yp;
        y[k] = y[k] + yp;
      }
    }
    half = span;
    span <<= 1;
    stride >>= 1;
  }
}

/* One forward fp32 FFT (interleaved re/im in and out). */
ORACLE_API int crn_oracle_fft_radix2(const float *x, float *y, int n) {
  plan_f32 p;
  if (n < 2 || plan_f32_init(&p, (unsigned)n)) return -1;
  plan_f32_exec(&p, (const float complex *)x, (float complex *)y);
  plan_f32_free(&p);
  return 0;
}

/* Float64 ground truth: O(N^2) DFT with long-double accumulation and exact angle reduction
 * (k*n mod N before the trig call).  Interleaved double in/out. */
ORACLE_API int crn_oracle_dft_f64(const double *x, double *y, int n) {
  if (n < 1) return -1;
  long double *c = (long double *)malloc(sizeof(long double) * 2 * (size_t)n);
  if (!c) return -1;
  for (int i = 0; i < n; i++) {
    long double a = -2.0L * 3.14159265358979323846264338327950288L * (long double)i / (long double)n;
    c[2 * i] = cosl(a);
    c[2 * i + 1] = sinl(a);
  }
  for (int k = 0; k < n; k++) {
    long double sr = 0, si = 0;
    for (int i = 0; i < n; i++) {
      const size_t idx = (size_t)(((long long)k * i) % n);
      const long double wr = c[2 * idx], wi = c[2 * idx + 1];
      const long double xr = x[2 * i], xi = x[2 * i + 1];
      sr += xr * wr - xi * wi;
      si += xr * wi + xi * wr;
    }
    y[2 * k] = (double)sr;
    y[2 * k + 1] = (double)si;
  }
  free(c);
  return 0;
}

/* ------------------------------------------------------------------------------------------
 * ANN forward + cascade — CE_Predictive_Node.cpp:200, 214-235, 245-261.
 * feat4 = {NOISE_FLOOR, CH1, CH2, CH3} (Features_Buffer[1..4]); out3 = Output[1..3];
 * returns 0 ("ALL BUSY": no output >= threshold) or the 1-based channel reported OCCUPIED.
 * ---------------------------------------------------------------------------------------- */
static int ann_decide(const double w_ih[5][6], const double w_ho[6][4], double thr,
                      const float feat4[4], double out3[3]) {
  double fb[5] = {0, (double)feat4[0], (double)feat4[1], (double)feat4[2], (double)feat4[3]};
  double hid[6] = {0};
  for (int j = 1; j <= 5; j++) {
    double s = w_ih[0][j];
    for (int i = 1; i <= 4; i++) s += fb[i] * w_ih[i][j];
    hid[j] = 1.0 / (1.0 + exp(-s));
  }
  double o[4] = {0};
  for (int k = 1; k <= 3; k++) {
    double s = w_ho[0][k];
    for (int j = 1; j <= 5; j++) s += hid[j] * w_ho[j][k];
    o[k] = 1.0 / (1.0 + exp(-s));
  }
  out3[0] = o[1];
  out3[1] = o[2];
  out3[2] = o[3];
  if (o[1] >= thr) return 1;
  if (o[2] >= thr) return 2;
  if (o[3] >= thr) return 3;
  return 0;
}

ORACLE_API int crn_oracle_ann(const float *feat4, double *out3) {
  return ann_decide(REF_W_IH, REF_W_HO, 0.8, feat4, out3);
}

/* ------------------------------------------------------------------------------------------
 * One reference epoch, literally: K = 10 packets of L samples each.
 *   iq        [10][L] interleaved complex fp32 — the successive contents of
 *             ECR->ce_usrp_rx_buffer on the 10 USRP_RX_SAMPS events of one epoch
 *   L         ce_usrp_rx_buffer_length; must be <= 512 here (the reference would overrun)
 *   fft_avg   [512] out: value of fft_avg[] just before the reset at .cpp:287
 *   feat4     out: NOISE_FLOOR, CH1, CH2, CH3 (.cpp:194-197)
 *   out3      out: Output[1..3]
 *   tx_freq   out: argument of the set_tx_freq call, or 0 when none is made (.cpp:260-261)
 * returns the decision 0..3, or -1 on bad arguments.
 * ---------------------------------------------------------------------------------------- */
ORACLE_API int crn_oracle_ref_epoch(const float *iq, int L, float *fft_avg, float *feat4,
                                    double *out3, double *tx_freq) {
  if (L < 1 || L > REF_N) return -1;
  plan_f32 plan;
  if (plan_f32_init(&plan, REF_N)) return -1;
  float complex buffer[REF_N];   /* .hpp:49, zeroed once at .cpp:37 */
  float complex buffer_F[REF_N]; /* .hpp:50 */
  float avg[REF_N];              /* .hpp:51, zeroed at .cpp:39 / :287 */
  memset(buffer, 0, sizeof(buffer));
  memset(buffer_F, 0, sizeof(buffer_F));
  memset(avg, 0, sizeof(avg));

  for (int f = 0; f < REF_K; f++) {
    memcpy(buffer, iq + (size_t)2 * f * L, (size_t)L * sizeof(float complex)); /* .cpp:149 */
    plan_f32_exec(&plan, buffer, buffer_F);                                      /* .cpp:150 */
    for (int i = 0; i < REF_N; i++)                                              /* .cpp:152-154 */
      avg[i] += cabsf(buffer_F[i]) / (float)REF_K;
  }
  plan_f32_free(&plan);

  /* .cpp:163-191: five ascending runs; cabsf(float) of a non-negative mean is the value itself */
  float m1 = 0.0f, m2 = 0.0f, m3 = 0.0f, nf = 0.0f;
  for (int i = 0; i < 16; i++) m1 += fabsf(avg[i]);
  for (int i = 496; i < 511; i++) m1 += fabsf(avg[i]); /* bin 511 is left out */
  for (int i = 55; i < 85; i++) m2 += fabsf(avg[i]);
  for (int i = 189; i < 222; i++) m3 += fabsf(avg[i]);
  for (int i = 300; i < 310; i++) nf += fabsf(avg[i]);
  const float ch1 = m1 * m1, ch2 = m2 * m2, ch3 = m3 * m3, noise = nf * nf; /* .cpp:194-197 */
  feat4[0] = noise;
  feat4[1] = ch1;
  feat4[2] = ch2;
  feat4[3] = ch3;
  if (fft_avg) memcpy(fft_avg, avg, sizeof(avg));

  const int d = ann_decide(REF_W_IH, REF_W_HO, 0.8, feat4, out3);
  if (tx_freq) {
    /* .cpp:245-258: CH1 busy -> CHANNEL2, CH2 busy -> CHANNEL1, CH3 busy -> CHANNEL2 */
    *tx_freq = d == 1 ? REF_CHANNEL2 : d == 2 ? REF_CHANNEL1 : d == 3 ? REF_CHANNEL2 : 0.0;
  }
  return d;
}

/* ------------------------------------------------------------------------------------------
 * Generalised pipeline (semantics documented in include/crn_sense.h).
 * ---------------------------------------------------------------------------------------- */
static int cfg_ok(const crn_cfg *c, int L) {
  if (!c || c->fft_len < 2 || c->frames_per_epoch < 1) return 0;
  if (c->n_bands < 1 || c->n_bands > CRN_MAX_BANDS) return 0;
  if (c->n_segs < 1 || c->n_segs > CRN_MAX_SEGS) return 0;
  if (L < 1 || L > c->fft_len) return 0;
  if (c->hop < 1 || c->hop > c->fft_len) return 0;
  if (c->hop != c->fft_len && L != c->fft_len) return 0;
  return 1;
}

static void process_epoch(const crn_cfg *c, const plan_f32 *plan, const float *win,
                          const float *iq_epoch, int L, float complex *x, float complex *X,
                          float *acc, float *features, double *ann_out, int32_t *decision,
                          uint8_t *occupancy, float *spectrum) {
  const int N = c->fft_len, K = c->frames_per_epoch;
  memset(acc, 0, sizeof(float) * (size_t)N);
  memset(x, 0, sizeof(float complex) * (size_t)N);
  for (int f = 0; f < K; f++) {
    const float complex *src = (const float complex *)(iq_epoch + (size_t)2 * f * (c->hop == N ? L : c->hop));
    if (win) {
      for (int i = 0; i < L; i++) x[i] = src[i] * win[i];
    } else {
      memcpy(x, src, (size_t)L * sizeof(float complex));
    }
    plan_f32_exec(plan, x, X);
    if (c->mode == CRN_MODE_REF_MAG) {
      for (int i = 0; i < N; i++) acc[i] += cabsf(X[i]) / (float)K;
    } else {
      for (int i = 0; i < N; i++) {
        const float re = crealf(X[i]), im = cimagf(X[i]);
        acc[i] += re * re + im * im;
      }
    }
  }
  if (c->mode == CRN_MODE_ENERGY)
    for (int i = 0; i < N; i++) acc[i] = acc[i] / (float)K;
  if (spectrum) memcpy(spectrum, acc, sizeof(float) * (size_t)N);

  float feat[CRN_MAX_BANDS];
  for (int b = 0; b < c->n_bands; b++) feat[b] = 0.0f;
  for (int s = 0; s < c->n_segs; s++) {
    const crn_band_seg *g = &c->segs[s];
    float m = feat[g->band];
    for (int i = g->lo; i < g->hi; i++) m += acc[i];
    feat[g->band] = m;
  }
  if (c->mode == CRN_MODE_REF_MAG)
    for (int b = 0; b < c->n_bands; b++) feat[b] = feat[b] * feat[b];
  if (features) memcpy(features, feat, sizeof(float) * (size_t)c->n_bands);

  if (c->decide == CRN_DECIDE_ANN) {
    double o[3];
    const int d = ann_decide(c->ann_w_ih, c->ann_w_ho, c->ann_threshold, feat, o);
    if (ann_out) memcpy(ann_out, o, sizeof(o));
    if (decision) *decision = d;
    if (occupancy)
      for (int b = 0; b < c->n_bands; b++) occupancy[b] = (uint8_t)(b >= 1 && b == d);
  } else if (c->decide == CRN_DECIDE_THRESHOLD) {
    const float ref = c->ref_band >= 0 ? feat[c->ref_band] : 1.0f;
    int cnt = 0;
    for (int b = 0; b < c->n_bands; b++) {
      const float t = c->thresh[b] * ref;
      const int occ = feat[b] > t;
      if (occupancy) occupancy[b] = (uint8_t)occ;
      cnt += occ;
    }
    if (decision) *decision = cnt;
  } else {
    if (decision) *decision = 0;
    if (occupancy) memset(occupancy, 0, (size_t)c->n_bands);
  }
}

/* Host-memory twin of crn_sense_run_host.  n_threads > 1 splits the epochs over OpenMP
 * threads (used only to time the "all host cores" CPU baseline). Returns 0 or -1. */
ORACLE_API int crn_oracle_run(const crn_cfg *c, const float *iq, int64_t n_epochs, int32_t L,
                              int64_t epoch_stride, const crn_out *out, int32_t n_threads) {
  if (!cfg_ok(c, L) || !iq || !out || n_epochs < 0) return -1;
  const int N = c->fft_len, K = c->frames_per_epoch;
  if (epoch_stride <= 0) epoch_stride = (int64_t)K * (c->hop == N ? L : c->hop);
  for (int s = 0; s < c->n_segs; s++) {
    const crn_band_seg *g = &c->segs[s];
    if (g->lo < 0 || g->hi > N || g->lo > g->hi || g->band < 0 || g->band >= c->n_bands) return -1;
  }
  if (c->decide == CRN_DECIDE_ANN && c->n_bands != 4) return -1;

  plan_f32 plan;
  if (plan_f32_init(&plan, (unsigned)N)) return -1;
  float *win = NULL;
  if (c->window == CRN_WINDOW_HANN) {
    win = (float *)malloc(sizeof(float) * (size_t)N);
    for (int i = 0; i < N; i++) win[i] = (float)(0.5 - 0.5 * cos(2.0 * M_PI * (double)i / (double)N));
  } else if (c->window == CRN_WINDOW_BLACKMAN_HARRIS) {
    /* spectrum_analyzer.py:262-275 (firdes.WIN_BLACKMAN_hARRIS): 4-term, symmetric over N-1 */
    win = (float *)malloc(sizeof(float) * (size_t)N);
    for (int i = 0; i < N; i++) {
      const double x = 2.0 * M_PI * (double)i / (double)(N - 1);
      win[i] = (float)(0.35875 - 0.48829 * cos(x) + 0.14128 * cos(2 * x) - 0.01168 * cos(3 * x));
    }
  }
  if (n_threads < 1) n_threads = 1;
  int fail = 0;
#pragma omp parallel num_threads(n_threads)
  {
    float complex *x = (float complex *)malloc(sizeof(float complex) * (size_t)N);
    float complex *X = (float complex *)malloc(sizeof(float complex) * (size_t)N);
    float *acc = (float *)malloc(sizeof(float) * (size_t)N);
    if (!x || !X || !acc) {
#pragma omp atomic write
      fail = 1;
    } else {
#pragma omp for schedule(static)
      for (int64_t e = 0; e < n_epochs; e++) {
        process_epoch(c, &plan, win, iq + 2 * e * epoch_stride, L, x, X, acc,
                      out->features ? out->features + e * c->n_bands : NULL,
                      out->ann_out ? out->ann_out + e * 3 : NULL,
                      out->decision ? out->decision + e : NULL,
                      out->occupancy ? out->occupancy + e * c->n_bands : NULL,
                      out->spectrum ? out->spectrum + e * N : NULL);
      }
    }
    free(x);
    free(X);
    free(acc);
  }
  free(win);
  plan_f32_free(&plan);
  return fail ? -1 : 0;
}

ORACLE_API int crn_oracle_abi_version(void) { return CRN_ABI_VERSION; }
