// crn_kernels.hip — gfx950 (MI355X) kernels of the spectrum-sensing hot path.
//
// One launch runs, for n_epochs decision epochs of K frames each, the whole of
//   cognitive_engines/CE_Predictive_Node/CE_Predictive_Node.cpp:148-261 (reference):
//   zero-padded staging (:149) -> forward DFT (:150, liquid-dsp fft_execute) -> per-bin
//   magnitude / energy running mean over K frames (:152-154) -> band sums (:173-191) ->
//   features (:194-200) -> 4-5-3 sigmoid net in double (:214-235) -> decision cascade (:245-261).
//
// Shape of the computation (DESIGN.md has the derivation):
//   N = 16 * 16 * R3 (R3 = 2, 4, 8, 16 -> N = 512 .. 4096).  A frame is handled by T = N/16
//   threads, 16 complex points per thread, in three register passes:
//     pass 1  radix-16 over r   : x[t + T r]            -> twiddle W_N^{t a}   -> LDS exchange 1
//     pass 2  radix-16 over m_hi: y_a[R3 m_hi + m_lo]   -> twiddle W_T^{m_lo c}-> LDS exchange 2
//     pass 3  radix-R3 over m_lo                        -> X[a + 16 c + 256 d] in registers
//   Exchange 2 stays inside the R3 lanes that share `a` (same wave), so a 4096-point frame needs
//   one s_barrier per frame (two with a single LDS buffer) and N <= 1024 needs none.
//   HBM is read exactly once: 8 B per input sample, coalesced 512 B per wave instruction.
//   |X|^2 (or |X|) is accumulated per bin in 16 registers per thread over the K frames; the band
//   reduction, the net and the cascade run once per epoch (epoch_close: from the registers for small
//   band plans, through an LDS image of the spectrum otherwise).
#include "crn_sense_impl.h"

namespace crn {

// Kernel forms selectable through crn_sense_set_variant (0 = default).  The shipped library (libcrnsense.so) compiles the two that are
// forms of the product — 13 (= 0, the default) and 2 (no pass-3 row pruning: what any band table outside the reference plan's rows runs
// anyway) — and refuses every other number.  libcrnsense_ab.so (-DCRN_AB_VARIANTS; tools/ and the A/B test) adds measurement forms:
// crn_dispatch_ab.h, the one seam in this file.  The numbers are the ones profiles/ and docs/history/ quote; the schedules, ablations
// and layouts of rounds 1-4 that were measured and not kept are gone from the tree (docs/history/removed_variants.md).
static constexpr int kNumVariants = 27, kDefaultVariant = 13;
#ifdef CRN_AB_VARIANTS
#include "crn_dispatch_ab.h"
#else
static bool measurement_variant(int) { return false; }           // no measurement forms in this build
static bool measurement_variant_traces(int) { return false; }
static void measurement_variant_desc(int, int *, int *) {}
template <int R3>
static bool launch_measurement_form(const SenseParams &, bool, bool, int, hipStream_t, hipError_t *) { return false; }
#endif

// Does this build of the library carry variant v?  (0 = default.)
bool sense_variant_available(int v) { return v == 0 || v == kDefaultVariant || v == 2 || measurement_variant(v); }
// ... and does it write time stamps over the ann_out buffer (so that the buffer must reach the kernel whatever the decision rule)?
bool sense_variant_traces(int v) { return measurement_variant_traces(v); }

// Wire-format input: crn_kernels_sc16.hip, linked only into a library built with `make SC16=1` (a weak reference: null when absent).
__attribute__((weak)) hipError_t launch_sense_sc16(const SenseParams &p, int fft_len, bool mag, bool win, int variant, hipStream_t stream, int *deal_rounds_run);

// The forms other than the default exist for N = 4096 only; other sizes always run the default.
template <int R3>
static hipError_t launch_r(const SenseParams &p, bool mag, bool win, int variant, hipStream_t stream, int *deal_rounds_run) {
  constexpr int kBase = kSpread | kLdsBlk | kPrioValu | kMulti;
  if constexpr (R3 <= 4) {   // a launch of a few epochs (crn_api.cpp sets deal_rounds): one epoch per workgroup, frames dealt to its lane groups
    if (p.deal_rounds > 0) {
      const hipError_t e = win ? launch_dealt_win<R3, 0>(p, stream) : launch_dealt<R3, 0>(p, mag, stream);
      if (e != hipErrorLaunchOutOfResources) {
        if (e == hipSuccess && deal_rounds_run) *deal_rounds_run = p.deal_rounds;
        return e;
      }
      SenseParams q = p;   // the device refused the LDS the frame slots need (crn_sense_kernel.h: launch_dealt_cfg): the streaming form takes it
      q.deal_rounds = 0;
      return launch_r<R3>(q, mag, win, variant, stream, deal_rounds_run);
    }
  }
  if (hipError_t e; launch_measurement_form<R3>(p, mag, win, variant, stream, &e)) return e;   // (libcrnsense_ab.so only)
  // Periodic Hann (the Welch configuration), whole frames, energy mode: the window rides in pass 1's first
  // butterflies and the first block of pass-2 twiddles is read ahead of its use (+1 % on the Welch stream, and 8
  // window registers fewer; the A/B numbers are in docs/history/DESIGN_r03.md §5)
  // (a windowed handle runs this whatever plain-kernel form it selects)
  if (win && !mag && p.hann_sym && p.L == Geo<R3>::N) {
    if constexpr (R3 == 16) {  // the Welch scan's plan (equal contiguous bands): band sums without the spectrum image
      if (p.aligned_shift != 0)
        return launch_cfg<Cfg<R3, 1, true, true, false, true, true, 3, true, true, kBase | kHannSym | kTw2Early | kAlignedBands>>(p, stream);
    }
    return launch_cfg<Cfg<R3, 1, true, true, false, true, true, 3, true, true, kBase | kHannSym | kTw2Early>>(p, stream);
  }
  // The plain 4096-point kernel runs 4 workgroups per CU with the compressed pass-1 table and pass 2
  // from LDS; everything else 3 per CU (windowed kernels carry 16 more registers: the window).
  // 3 workgroups per CU, all 30 twiddles in registers at N < 4096 (4 per CU with the compressed tables was
  // measured at these sizes: equal at 1024, -3 % at 512, -7 % at 2048), streaming workgroups
  // (+2.5-3 % everywhere; N = 1024 used to spill with them until the epoch close was slimmed).
  // Windowed kernels (16 window registers, and for Welch three half-frame sets) read the pass-2
  // twiddles from LDS at every size: in registers they spill inside the frame loop.
  if (win) return launch_default<R3, 1, true, true, true, 3, true, kBase, 2>(p, mag, win, stream);
  if (R3 != 16 || mag || p.L != Geo<R3>::N) return launch_default<R3, 1, true, true, false, 3, true, kBase, 1, true, R3 != 16>(p, mag, win, stream);
  if constexpr (R3 == 16) {
    constexpr int kPlain = kSpread | kLdsBlk | kTw1C | kMulti | kPrioValu;
    const bool regb = reg_bands(p);
    if (variant != 2 && regb && (p.acc_mask & ~kRefPlanRows) == 0)   // the reference channel plan's rows only (the default, 13)
      return launch_rn<R3, 1, true, true, true, 4, true, kPlain | kRows | kRegBands>(p, mag, win, stream);
    // another plan, a per-bin spectrum request, or variant 2: no pruning
    if (regb) return launch_rn<R3, 1, true, true, true, 4, true, kPlain | kRegBands>(p, mag, win, stream);
    return launch_rn<R3, 1, true, true, true, 4, true, kPlain>(p, mag, win, stream);
  }
  return hipErrorInvalidValue;
}

hipError_t launch_sense(const SenseParams &p, int fft_len, bool mag, bool win, int variant,
                        hipStream_t stream, bool sc16, int *deal_rounds_run) {
  if (deal_rounds_run) *deal_rounds_run = 0;
  if (sc16) return launch_sense_sc16 ? launch_sense_sc16(p, fft_len, mag, win, variant, stream, deal_rounds_run) : hipErrorNotSupported;
  switch (fft_len) {
    case 512: return launch_r<2>(p, mag, win, variant, stream, deal_rounds_run);
    case 1024: return launch_r<4>(p, mag, win, variant, stream, deal_rounds_run);
    case 2048: return launch_r<8>(p, mag, win, variant, stream, deal_rounds_run);
    case 4096: return launch_r<16>(p, mag, win, variant, stream, deal_rounds_run);
    default: return hipErrorInvalidValue;
  }
}

int sense_num_variants() { return kNumVariants; }

int sense_deal_rounds(int fft_len, bool mag, bool win, bool hann_whole_frames, int K, size_t lds_budget) {
  if (fft_len != 512 && fft_len != 1024) return 0;
  if (win && (mag || !hann_whole_frames)) return 0;   // the one windowed dealt form: periodic Hann, energy mode, whole frames
  const int r3 = fft_len / 256, t = 16 * r3, groups = 256 / t;
  if (K < 2) return 0;   // one frame: nothing to deal
  const int rounds = (K + groups - 1) / groups;
  const size_t lds = ((size_t)groups * 16 * (t + r3) + 16 * r3) * sizeof(cx) + kCloseLdsBytes + (size_t)rounds * groups * fft_len * (mag ? 4 : 8);
  return lds <= lds_budget ? rounds : 0;
}
unsigned sense_ref_acc_mask(int fft_len) { return ref_acc_mask(fft_len / 256); }

void sense_variant(int fft_len, int variant, int *nbuf, int *prefetch, int *nt, int *tw2lds, int *pk) {
  *nbuf = 1; *prefetch = 1; *nt = 1; *pk = 1;
  *tw2lds = (variant >= 0 && fft_len == 4096) ? 1 : 0;   // the plain 4096-point kernel reads its pass-2 twiddles from LDS (the other
                                                         // sizes' plain kernels keep them in registers; windowed kernels: crn_api.cpp)
  if (fft_len == 4096 && measurement_variant(variant)) measurement_variant_desc(variant, nbuf, tw2lds);
}

void sense_geometry(int fft_len, int variant, int *threads, int *lds_bytes, int *epochs_per_block) {
  const int r3 = fft_len / 256;
  const int t = 16 * r3;
  const int groups = 256 / t;
  int nbuf, pf, nt, tl, pk;
  sense_variant(fft_len, variant, &nbuf, &pf, &nt, &tl, &pk);
  *threads = 256;
  *epochs_per_block = groups;
  (void)tl;
  *lds_bytes = (groups * nbuf * 16 * (t + r3) + 16 * r3) * 8 + kCloseLdsBytes;
}

// ---------------------------------------------------------------------------------------------
// Plain batched forward FFT (complex in, complex out): the transform of sense_kernel on its own, for
// callers that bind the liquid-dsp entry points the reference calls (include/crn_liquid_fft.h;
// CE_Predictive_Node.cpp:42-45,150).  One frame per T = N/16 threads, the same three passes and two
// exchanges; the spectrum goes back through the exchange buffer in natural order so the global
// stores are coalesced.  Not the hot path: no prefetch, no streaming.
// ---------------------------------------------------------------------------------------------
template <int R3>
__global__ __launch_bounds__(256, 2) void fft_kernel(const FftParams p) {
  using C = Cfg<R3, 1, false, false, false, false, false, 2, false, true, kLdsBlk>;
  using G = Geo<R3>;
  constexpr int T = G::T, N = G::N, J = G::J;
  extern __shared__ __attribute__((aligned(16))) cx lds[];
  const int tid = threadIdx.x;
  const int grp = tid / T, t = tid % T;
  FrameCtx<C> c;
  c.t = t;
  c.a = t / R3;
  c.m_lo = t % R3;
  c.L = p.L;
  c.gbuf = lds + grp * G::GROUP_CPLX;
  c.tw2_lds = nullptr;
#pragma unroll
  for (int i = 1; i < 16; i++) c.tw1[i] = reinterpret_cast<const cx *>(p.tw1)[i * T + t];
#pragma unroll
  for (int i = 1; i < 16; i++) c.tw2[i] = reinterpret_cast<const cx *>(p.tw2)[i * R3 + c.m_lo];
  const long long fr = (long long)blockIdx.x * G::GROUPS + grp;
  const bool live = fr < p.n_frames;
  cx u[16], v[16];
#pragma unroll
  for (int r = 0; r < 16; r++) {
    const int n = t + T * r;
    u[r] = (live && n < p.L) ? reinterpret_cast<const cx *>(p.in)[fr * p.frame_stride + n] : cx{0.f, 0.f};
  }
  ph_pass1<C>(u, v, c);
  ph_x1_write<C>(v, c.gbuf, c);
  group_sync<C>();
  ph_x1_read<C>(u, c.gbuf, c);
  ph_pass2<C>(u, v, c);
  wave_sync();
  ph_x2_write<C>(v, c.gbuf, c);
  wave_sync();
  ph_x2_read<C>(u, c.gbuf, c);
  ph_pass3<C>(u, v);
  group_sync<C>();  // every wave of the group is done with the exchange buffer
#pragma unroll
  for (int j = 0; j < J; j++)
#pragma unroll
    for (int d = 0; d < R3; d++) {
      const int k = c.a + 16 * (c.m_lo * J + j) + 256 * d;
      c.gbuf[spec_phys(k)] = v[j * R3 + d];
    }
  group_sync<C>();
  if (live) {
    cx *dst = reinterpret_cast<cx *>(p.out) + fr * N;
#pragma unroll
    for (int r = 0; r < 16; r++) dst[t + T * r] = c.gbuf[spec_phys(t + T * r)];
  }
}

template <int R3>
static hipError_t launch_fft_r(const FftParams &p, hipStream_t stream) {
  using G = Geo<R3>;
  const long long grid = (p.n_frames + G::GROUPS - 1) / G::GROUPS;
  if (grid <= 0) return hipSuccess;
  const size_t lds = (size_t)G::GROUPS * G::GROUP_CPLX * sizeof(cx);
  hipLaunchKernelGGL(fft_kernel<R3>, dim3((unsigned)grid), dim3(256), lds, stream, p);
  return hipGetLastError();
}

hipError_t launch_fft(const FftParams &p, int fft_len, hipStream_t stream) {
  switch (fft_len) {
    case 512: return launch_fft_r<2>(p, stream);
    case 1024: return launch_fft_r<4>(p, stream);
    case 2048: return launch_fft_r<8>(p, stream);
    case 4096: return launch_fft_r<16>(p, stream);
    default: return hipErrorInvalidValue;
  }
}

// ---------------------------------------------------------------------------------------------
// synthetic IQ generator (measurement / test aid; see crn_synth_fill_device in crn_sense.h).
// Mirrors the traffic shape of cognitive_engines/CE_Random_Behaviour_PU (a PU hopping uniformly
// over the channels, CE_Random_Behaviour_PU.cpp:41-53) as seeded on-grid tones over AWGN.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t mix64(uint64_t z) {  // splitmix64 finaliser
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

// ---- modulated carriers (CRN_SIG_RRC_QPSK / _GMSK / _OFDM, include/crn_sense.h).  Double precision on purpose: this is the test
// signal source, not the sensing path, and the pulse formulas are 0/0 forms next to their removable singularities.
constexpr double kPiD = 3.14159265358979323846;
constexpr double kRrcBeta = 0.35;  // RRC_BETA, include/interferer.hpp:21
constexpr int kRrcSemi = 8;        // symbols either side kept of the pulse

__device__ __forceinline__ double rrc_pulse(double tau) {  // unit-energy root-raised-cosine, symbol period 1
  const double q = 4.0 * kRrcBeta * tau;
  if (fabs(tau) < 1e-9) return 1.0 - kRrcBeta + 4.0 * kRrcBeta / kPiD;
  if (fabs(fabs(q) - 1.0) < 1e-9)
    return kRrcBeta / sqrt(2.0) * ((1.0 + 2.0 / kPiD) * sin(kPiD / (4.0 * kRrcBeta)) + (1.0 - 2.0 / kPiD) * cos(kPiD / (4.0 * kRrcBeta)));
  return (sin(kPiD * tau * (1.0 - kRrcBeta)) + q * cos(kPiD * tau * (1.0 + kRrcBeta))) / (kPiD * tau * (1.0 - q * q));
}

// GMSK phase pulse, BT = 0.5: the integral of a unit rectangle smoothed by a Gaussian of sigma = sqrt(ln 2) / (2 pi BT) symbols;
// 0 well before the symbol, 1 well after.  F(x) = x Phi(x / sigma) + sigma phi(x / sigma) is the antiderivative of Phi(x / sigma).
__device__ __forceinline__ double gmsk_ramp(double x) {
  const double sigma = 0.26501095104247255;  // sqrt(ln 2) / pi
  const double z = x / sigma;
  return x * 0.5 * erfc(-z * 0.70710678118654752440) + sigma * 0.39894228040143267794 * exp(-0.5 * z * z);
}
__device__ __forceinline__ double gmsk_phase_pulse(double tau) { return gmsk_ramp(tau + 0.5) - gmsk_ramp(tau - 0.5); }

// data bits of a GMSK burst: 64 per hash word, so that the running sum of +-1 up to any symbol costs one popcount per word
__device__ __forceinline__ uint64_t gmsk_word(uint64_t hsig, long long j) { return mix64(hsig ^ mix64(0x6D5Bull + (uint64_t)j)); }
__device__ __forceinline__ long long gmsk_prefix(uint64_t hsig, long long kk) {  // sum of b[0..kk], b = +-1
  long long sum = 0;
  const long long blk = kk >> 6;
  for (long long j = 0; j < blk; j++) sum += 2 * __popcll(gmsk_word(hsig, j)) - 64;
  const int cnt = (int)(kk & 63) + 1;
  const uint64_t mask = cnt == 64 ? ~0ull : ((1ull << cnt) - 1ull);
  return sum + 2 * __popcll(gmsk_word(hsig, blk) & mask) - cnt;
}

// One sample (index m of its epoch) of the modulated carrier of `kind` filling a band of nb bins on an fft_len grid, before the
// carrier: unit power.
__device__ void modulated_baseband(int kind, uint64_t hsig, long long m, int nb, int fft_len, double *out_re, double *out_im) {
  double re = 0.0, im = 0.0;
  if (kind == 3) {  // RRC QPSK: (1 + beta) Rs = band width
    const double sps = (double)fft_len * (1.0 + kRrcBeta) / (double)nb;
    const double tau0 = (double)m / sps;
    const long long k0 = (long long)floor(tau0);
    for (long long k = k0 - kRrcSemi + 1; k <= k0 + kRrcSemi; k++) {
      const uint64_t hs = mix64(hsig ^ mix64(0x5EEDull + (uint64_t)(k + 64)));
      const double hv = rrc_pulse(tau0 - (double)k) * 0.70710678118654752440;
      re += (hs & 1ull) ? hv : -hv;
      im += (hs & 2ull) ? hv : -hv;
    }
  } else if (kind == 4) {  // GMSK: phase = pi / 2 * sum_k b_k q(t / T - k)
    const double sps = 1.5 * (double)fft_len / (double)nb;
    const double tau0 = (double)m / sps;
    const long long k0 = (long long)floor(tau0);
    double acc = (double)(gmsk_prefix(hsig, k0 - 3 + 8) & 3);   // symbols that have fully turned, modulo a full circle
    for (long long k = k0 - 2; k <= k0 + 3; k++) {
      const long long kk = k + 8;
      const double b = ((gmsk_word(hsig, kk >> 6) >> (kk & 63)) & 1ull) ? 1.0 : -1.0;
      acc += b * gmsk_phase_pulse(tau0 - (double)k);
    }
    re = cos(0.5 * kPiD * acc);
    im = sin(0.5 * kPiD * acc);
  } else {  // OFDM: subcarriers 15 kHz apart at 13 MHz, cyclic prefix 1/4
    const double d = 15.0e3 / 13.0e6 * (double)fft_len;   // spacing in bins
    const double tu = (double)fft_len / d, ts = 1.25 * tu;
    int nsub = (int)floor((double)nb / d);
    if (nsub < 1) nsub = 1;
    const long long q = (long long)floor((double)m / ts);
    const double t_in = (double)m - (double)q * ts - 0.25 * tu;
    const double amp = 1.0 / sqrt(2.0 * (double)nsub);
    for (int i = 0; i < nsub; i++) {
      const uint64_t hs = mix64(hsig ^ mix64(0xFD0000000000ull + ((uint64_t)q << 20) + (uint64_t)i));
      const double turns = ((double)i - 0.5 * (double)(nsub - 1)) * d * t_in / (double)fft_len;
      const double a = 2.0 * kPiD * (turns - floor(turns));
      const double c = cos(a), sn = sin(a);
      const double ar = (hs & 1ull) ? amp : -amp, ai = (hs & 2ull) ? amp : -amp;
      re += ar * c - ai * sn;
      im += ar * sn + ai * c;
    }
  }
  *out_re = re;
  *out_im = im;
}

__global__ __launch_bounds__(256) void synth_kernel(const SynthParams p) {
  const long long total = p.n_epochs * p.samples_per_epoch;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const long long e = i / p.samples_per_epoch;
    const long long n = i - e * p.samples_per_epoch;
    const uint64_t h = mix64(p.seed ^ mix64((uint64_t)i));
    const float u1 = ((float)(uint32_t)(h >> 40) + 0.5f) * (1.0f / 16777216.0f);
    const float u2 = ((float)(uint32_t)((h >> 16) & 0xFFFFFF)) * (1.0f / 16777216.0f);
    const float r = p.noise_sigma * sqrtf(-2.0f * logf(u1));
    float sn, cs;
    sincospif(2.0f * u2, &sn, &cs);
    float re = r * cs, im = r * sn;

    const uint64_t he = mix64(p.seed * 0x9E3779B97F4A7C15ull + (uint64_t)e + 0x51ED27ull);
    // uniform model: independent pick per epoch (0 = idle); Markov models: state written by pu_pattern_kernel
    int pick;
    if (p.pu_model == 0) {
      pick = (int)(he % (uint64_t)(p.n_active + 1));
    } else if (p.pu_model == 3) {  // sweep: up one band per epoch, reverse at either end (src/interferer.cpp:339-345)
      const int period = 2 * (p.n_active - 1);
      const int pos = period > 0 ? (int)((e % p.epochs_per_stream) % period) : 0;
      pick = 1 + (pos < p.n_active ? pos : period - pos);
    } else {
      pick = p.truth[e];
    }
    if (pick > 0 && p.signal_kind >= 3) {
      const int band = p.active_band0 + pick - 1;
      const int nb = p.band_bins_begin[band + 1] - p.band_bins_begin[band];
      double br, bi;
      modulated_baseband(p.signal_kind, he, n, nb, p.fft_len, &br, &bi);
      const long long two_n = 2ll * p.fft_len;
      const long long c2 = ((long long)p.band_c2[band] % two_n + two_n) % two_n;
      const double a = 2.0 * kPiD * (double)((c2 * (n % two_n)) % two_n) / (double)two_n;
      const double c = cos(a), sn = sin(a);
      re += (float)((double)p.signal_rms * (br * c - bi * sn));
      im += (float)((double)p.signal_rms * (br * sn + bi * c));
    } else if (pick > 0) {
      const int band = p.active_band0 + pick - 1;
      const int nb = p.band_bins_begin[band + 1] - p.band_bins_begin[band];
      const int *bins = p.band_bins + p.band_bins_begin[band];
      const int nmod = (int)(n % p.fft_len);
      int nt = p.tones < nb ? p.tones : nb;
      float amp = p.tone_amp;
      uint64_t hsig = he;
      if (p.signal_kind == 1) {  // CW: one carrier at the band centre
        nt = 1;
        amp = p.signal_rms;
      } else if (p.signal_kind == 2) {  // every bin, new phases each frame
        nt = nb;
        amp = p.signal_rms * rsqrtf((float)nb);
        hsig = mix64(he ^ (0xF00Dull + (uint64_t)(n / p.fft_len)));
      }
      for (int j = 0; j < nt; j++) {
        const int k = p.signal_kind == 2 ? bins[j] : bins[(int)(((long long)(2 * j + 1) * nb) / (2 * nt))];
        const uint64_t hp = mix64(hsig + 0x1234567ull * (uint64_t)(j + 1));
        const float phase2 = (float)(uint32_t)(hp >> 40) * (2.0f / 16777216.0f);  // in units of pi
        const int kn = (int)(((long long)k * nmod) % p.fft_len);
        float s, c;
        sincospif(2.0f * (float)kn / (float)p.fft_len + phase2, &s, &c);
        re = fmaf(amp, c, re);
        im = fmaf(amp, s, im);
      }
    }
    if (p.adc_scale > 0.f) {  // the radio's integer samples as UHD hands them over
      re = fminf(fmaxf(rintf(re * p.adc_scale), -p.adc_scale), p.adc_scale - 1.f) / p.adc_scale;
      im = fminf(fmaxf(rintf(im * p.adc_scale), -p.adc_scale), p.adc_scale - 1.f) / p.adc_scale;
    }
    p.iq[i] = make_float2(re, im);
    if (n == 0 && p.truth != nullptr && (p.pu_model == 0 || p.pu_model == 3)) p.truth[e] = pick;
  }
}

// Markov traffic models (include/crn_sense.h, crn_pu_model): one thread walks one stream's chain.
// States 1..3 = CH1..CH3; the outcome of step j of stream s is a counter hash mod 10, standing in
// for the reference's rand() % 10 (CE_PU_MARKOV_Chain_Tx.cpp:82-86).
__global__ __launch_bounds__(64) void pu_pattern_kernel(const SynthParams p) {
  const long long n_streams = p.n_epochs / p.epochs_per_stream;
  const long long s = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n_streams) return;
  int state = 1;
  for (long long j = 0; j < p.epochs_per_stream; j++) {
    const int outcome = (int)(mix64(p.seed ^ mix64(0xA5A5A5A5ull + (uint64_t)s * 0x100000001B3ull + (uint64_t)j)) % 10ull);
    int next;
    if (p.pu_model == 1) {  // as written: `>= 1 || < 4` is always true
      next = outcome == 0 ? 1 : 2;
    } else {                // as intended
      const int stay2 = state == 2 ? 5 : 3;  // CH2 keeps outcomes 1..5, the others send 1..3 to CH2
      next = outcome == 0 ? 1 : outcome <= stay2 ? 2 : 3;
    }
    state = next > p.n_active ? p.n_active : next;  // plans with fewer than three driven bands
    p.truth[s * p.epochs_per_stream + j] = state;
  }
}

hipError_t launch_pu_pattern(const SynthParams &p, hipStream_t stream) {
  if (p.n_epochs <= 0 || p.epochs_per_stream <= 0) return hipSuccess;
  const long long n_streams = p.n_epochs / p.epochs_per_stream;
  hipLaunchKernelGGL(pu_pattern_kernel, dim3((unsigned)((n_streams + 63) / 64)), dim3(64), 0, stream, p);
  return hipGetLastError();
}

__global__ void nop_kernel() {}
hipError_t launch_nop(hipStream_t stream) {
  hipLaunchKernelGGL(nop_kernel, dim3(1), dim3(64), 0, stream);
  return hipGetLastError();
}

hipError_t launch_synth(const SynthParams &p, hipStream_t stream) {
  const long long total = p.n_epochs * p.samples_per_epoch;
  if (total <= 0) return hipSuccess;
  long long blocks = (total + 255) / 256;
  if (blocks > 256 * 32) blocks = 256 * 32;
  hipLaunchKernelGGL(synth_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, p);
  return hipGetLastError();
}

}  // namespace crn
