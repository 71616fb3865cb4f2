/* crn_oracle_synth.c — TEST INFRASTRUCTURE ONLY (see crn_oracle.c's header): CPU twin of the device
 * signal generator (crn_synth_fill_device_ex, csrc/crn_kernels.hip synth_kernel / pu_pattern_kernel).
 *
 * The generator has no counterpart in the reference to be exact against — it is this build's
 * measurement aid, shaped after the reference's traffic sources:
 *   uniform channel pick ......... cognitive_engines/CE_Random_Behaviour_PU/CE_Random_Behaviour_PU.cpp:41-53
 *   Markov chain, as written ..... cognitive_engines/CE_PU_MARKOV_Chain_Tx/CE_PU_MARKOV_Chain_Tx.cpp:82-128
 *   CW / noise / multicarrier .... src/interferer.cpp:128-140, 248-282
 *   frequency sweep .............. src/interferer.cpp:339-345
 *   RRC QPSK / GMSK / OFDM ....... src/interferer.cpp:160-282 (the waveforms liquid-dsp's frame generators make there,
 *                                  restated from their textbook definitions: see include/crn_sense.h, crn_signal_kind)
 * The twin restates the same counter hashes and formulas in plain C so that tests can check the
 * device kernel sample by sample (fp32 transcendentals differ by an ulp or two between libm and
 * the device library: tests allow 1e-5 of the signal scale) and the occupancy truth exactly.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#include "../include/crn_sense.h"

static uint64_t mix64(uint64_t z) { /* splitmix64 step: add the golden-ratio increment, then finalise */
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

/* ---- modulated carriers: same formulas, constants and hash keys as csrc/crn_kernels.hip (modulated_baseband) ---- */
#define RRC_BETA 0.35
#define RRC_SEMI 8

static double rrc_pulse(double tau) {
  const double q = 4.0 * RRC_BETA * tau;
  if (fabs(tau) < 1e-9) return 1.0 - RRC_BETA + 4.0 * RRC_BETA / M_PI;
  if (fabs(fabs(q) - 1.0) < 1e-9)
    return RRC_BETA / sqrt(2.0) * ((1.0 + 2.0 / M_PI) * sin(M_PI / (4.0 * RRC_BETA)) + (1.0 - 2.0 / M_PI) * cos(M_PI / (4.0 * RRC_BETA)));
  return (sin(M_PI * tau * (1.0 - RRC_BETA)) + q * cos(M_PI * tau * (1.0 + RRC_BETA))) / (M_PI * tau * (1.0 - q * q));
}
static double gmsk_ramp(double x) {
  const double sigma = 0.26501095104247255; /* sqrt(ln 2) / (2 pi BT), BT = 0.5 */
  const double z = x / sigma;
  return x * 0.5 * erfc(-z * M_SQRT1_2) + sigma * 0.39894228040143267794 * exp(-0.5 * z * z);
}
static double gmsk_phase_pulse(double tau) { return gmsk_ramp(tau + 0.5) - gmsk_ramp(tau - 0.5); }
static uint64_t gmsk_word(uint64_t hsig, int64_t j) { return mix64(hsig ^ mix64(0x6D5Bull + (uint64_t)j)); }
static int64_t gmsk_prefix(uint64_t hsig, int64_t kk) {
  int64_t sum = 0;
  const int64_t blk = kk >> 6;
  for (int64_t j = 0; j < blk; j++) sum += 2 * __builtin_popcountll(gmsk_word(hsig, j)) - 64;
  const int cnt = (int)(kk & 63) + 1;
  const uint64_t mask = cnt == 64 ? ~0ull : ((1ull << cnt) - 1ull);
  return sum + 2 * __builtin_popcountll(gmsk_word(hsig, blk) & mask) - cnt;
}
static void modulated_baseband(int kind, uint64_t hsig, int64_t m, int nb, int fft_len, double *out_re, double *out_im) {
  double re = 0.0, im = 0.0;
  if (kind == CRN_SIG_RRC_QPSK) {
    const double sps = (double)fft_len * (1.0 + RRC_BETA) / (double)nb;
    const double tau0 = (double)m / sps;
    const int64_t k0 = (int64_t)floor(tau0);
    for (int64_t k = k0 - RRC_SEMI + 1; k <= k0 + RRC_SEMI; k++) {
      const uint64_t hs = mix64(hsig ^ mix64(0x5EEDull + (uint64_t)(k + 64)));
      const double hv = rrc_pulse(tau0 - (double)k) * M_SQRT1_2;
      re += (hs & 1ull) ? hv : -hv;
      im += (hs & 2ull) ? hv : -hv;
    }
  } else if (kind == CRN_SIG_GMSK) {
    const double sps = 1.5 * (double)fft_len / (double)nb;
    const double tau0 = (double)m / sps;
    const int64_t k0 = (int64_t)floor(tau0);
    double acc = (double)(gmsk_prefix(hsig, k0 - 3 + 8) & 3);
    for (int64_t k = k0 - 2; k <= k0 + 3; k++) {
      const int64_t kk = k + 8;
      const double b = ((gmsk_word(hsig, kk >> 6) >> (kk & 63)) & 1ull) ? 1.0 : -1.0;
      acc += b * gmsk_phase_pulse(tau0 - (double)k);
    }
    re = cos(0.5 * M_PI * acc);
    im = sin(0.5 * M_PI * acc);
  } else {
    const double d = 15.0e3 / 13.0e6 * (double)fft_len;
    const double tu = (double)fft_len / d, ts = 1.25 * tu;
    int nsub = (int)floor((double)nb / d);
    if (nsub < 1) nsub = 1;
    const int64_t q = (int64_t)floor((double)m / ts);
    const double t_in = (double)m - (double)q * ts - 0.25 * tu;
    const double amp = 1.0 / sqrt(2.0 * (double)nsub);
    for (int i = 0; i < nsub; i++) {
      const uint64_t hs = mix64(hsig ^ mix64(0xFD0000000000ull + ((uint64_t)q << 20) + (uint64_t)i));
      const double turns = ((double)i - 0.5 * (double)(nsub - 1)) * d * t_in / (double)fft_len;
      const double a = 2.0 * M_PI * (turns - floor(turns));
      const double c = cos(a), sn = sin(a);
      const double ar = (hs & 1ull) ? amp : -amp, ai = (hs & 2ull) ? amp : -amp;
      re += ar * c - ai * sn;
      im += ar * sn + ai * c;
    }
  }
  *out_re = re;
  *out_im = im;
}

/* Chain step of CE_PU_MARKOV_Chain_Tx::PU_TX_Behaviour (.cpp:92-127); states 1..3 = CH1..CH3. */
static int markov_next(int model, int state, int outcome) {
  if (model == CRN_PU_MARKOV_AS_WRITTEN) return outcome == 0 ? 1 : 2; /* `>= 1 || < 4`: always true */
  const int stay2 = state == 2 ? 5 : 3;
  return outcome == 0 ? 1 : outcome <= stay2 ? 2 : 3;
}

__attribute__((visibility("default")))
int crn_oracle_synth(const crn_cfg *c, const crn_synth_cfg *sc, float *iq, int64_t n_epochs,
                     int64_t spe, int32_t *truth) {
  if (!c || !sc || !iq || !truth || n_epochs < 0 || spe < 1 || sc->n_streams < 1) return -1;
  if (sc->adc_bits != 0 && (sc->adc_bits < 2 || sc->adc_bits > 24)) return -1;
  /* bins of every band, table order inside a band (as crn_sense_create lays them out) */
  int *begin = (int *)calloc((size_t)c->n_bands + 1, sizeof(int));
  int total = 0;
  for (int s = 0; s < c->n_segs; s++) total += c->segs[s].hi - c->segs[s].lo;
  int *bins = (int *)malloc(sizeof(int) * (size_t)(total > 0 ? total : 1));
  int n = 0;
  for (int b = 0; b < c->n_bands; b++) {
    begin[b] = n;
    for (int s = 0; s < c->n_segs; s++)
      if (c->segs[s].band == b)
        for (int k = c->segs[s].lo; k < c->segs[s].hi; k++) bins[n++] = k;
  }
  begin[c->n_bands] = n;
  int active0, n_active;
  if (c->ref_band >= 0 || c->decide == CRN_DECIDE_ANN) {
    active0 = 1;
    n_active = c->n_bands - 1 < 3 ? c->n_bands - 1 : 3;
  } else {
    active0 = 0;
    n_active = c->n_bands;
  }
  if (sc->signal_kind == CRN_SIG_TONES && sc->tones_per_band == 0) n_active = 0;
  const float sigma = sqrtf(sc->noise_power * 0.5f);
  const float tone_amp = sc->tones_per_band > 0 ? sc->signal_rms / sqrtf((float)sc->tones_per_band) : 0.f;

  /* twice the signed centre of every band: lowest + highest signed bin (carrier of the modulated kinds) */
  int *c2 = (int *)calloc((size_t)c->n_bands + 1, sizeof(int));
  for (int b = 0; b < c->n_bands; b++) {
    int lo = c->fft_len, hi = -c->fft_len;
    for (int i = begin[b]; i < begin[b + 1]; i++) {
      const int k = bins[i] >= c->fft_len / 2 ? bins[i] - c->fft_len : bins[i];
      if (k < lo) lo = k;
      if (k > hi) hi = k;
    }
    c2[b] = begin[b + 1] > begin[b] ? lo + hi : 0;
  }

  /* occupancy pattern */
  if (sc->pu_model == CRN_PU_SWEEP) {
    if (n_epochs % sc->n_streams != 0 || n_active < 1) { free(begin); free(bins); free(c2); return -1; }
    const int64_t eps = n_epochs / sc->n_streams;
    const int period = 2 * (n_active - 1);
    for (int64_t e = 0; e < n_epochs; e++) {
      const int pos = period > 0 ? (int)((e % eps) % period) : 0;
      truth[e] = 1 + (pos < n_active ? pos : period - pos);
    }
  } else if (sc->pu_model != CRN_PU_UNIFORM) {
    if (n_epochs % sc->n_streams != 0 || n_active < 1) { free(begin); free(bins); free(c2); return -1; }
    const int64_t eps = n_epochs / sc->n_streams;
    for (int64_t s = 0; s < sc->n_streams; s++) {
      int state = 1;
      for (int64_t j = 0; j < eps; j++) {
        const int outcome = (int)(mix64(sc->seed ^ mix64(0xA5A5A5A5ull + (uint64_t)s * 0x100000001B3ull + (uint64_t)j)) % 10ull);
        const int next = markov_next(sc->pu_model, state, outcome);
        state = next > n_active ? n_active : next;
        truth[s * eps + j] = state;
      }
    }
  }

  for (int64_t e = 0; e < n_epochs; e++) {
    const uint64_t he = mix64(sc->seed * 0x9E3779B97F4A7C15ull + (uint64_t)e + 0x51ED27ull);
    const int pick = sc->pu_model == CRN_PU_UNIFORM ? (int)(he % (uint64_t)(n_active + 1)) : truth[e];
    if (sc->pu_model == CRN_PU_UNIFORM) truth[e] = pick;
    for (int64_t m = 0; m < spe; m++) {
      const int64_t i = e * spe + m;
      const uint64_t h = mix64(sc->seed ^ mix64((uint64_t)i));
      const float u1 = ((float)(uint32_t)(h >> 40) + 0.5f) * (1.0f / 16777216.0f);
      const float u2 = ((float)(uint32_t)((h >> 16) & 0xFFFFFF)) * (1.0f / 16777216.0f);
      const float r = sigma * sqrtf(-2.0f * logf(u1));
      const double a0 = 2.0 * M_PI * (double)u2;
      float re = r * (float)cos(a0), im = r * (float)sin(a0);
      if (pick > 0 && sc->signal_kind >= CRN_SIG_RRC_QPSK) {
        const int band = active0 + pick - 1;
        const int nb = begin[band + 1] - begin[band];
        double br, bi;
        modulated_baseband(sc->signal_kind, he, m, nb, c->fft_len, &br, &bi);
        const long long two_n = 2ll * c->fft_len;
        const long long cc = ((long long)c2[band] % two_n + two_n) % two_n;
        const double a = 2.0 * M_PI * (double)((cc * (m % two_n)) % two_n) / (double)two_n;
        const double cs = cos(a), sn = sin(a);
        re += (float)((double)sc->signal_rms * (br * cs - bi * sn));
        im += (float)((double)sc->signal_rms * (br * sn + bi * cs));
      } else if (pick > 0) {
        const int band = active0 + pick - 1;
        const int nb = begin[band + 1] - begin[band];
        const int *bb = bins + begin[band];
        const int nmod = (int)(m % c->fft_len);
        int nt = sc->tones_per_band < nb ? sc->tones_per_band : nb;
        float amp = tone_amp;
        uint64_t hsig = he;
        if (sc->signal_kind == CRN_SIG_CW) {
          nt = 1;
          amp = sc->signal_rms;
        } else if (sc->signal_kind == CRN_SIG_BAND_NOISE) {
          nt = nb;
          amp = sc->signal_rms / sqrtf((float)nb);
          hsig = mix64(he ^ (0xF00Dull + (uint64_t)(m / c->fft_len)));
        }
        for (int j = 0; j < nt; j++) {
          const int k = sc->signal_kind == CRN_SIG_BAND_NOISE ? bb[j] : bb[(int)(((long long)(2 * j + 1) * nb) / (2 * nt))];
          const uint64_t hp = mix64(hsig + 0x1234567ull * (uint64_t)(j + 1));
          const float phase2 = (float)(uint32_t)(hp >> 40) * (2.0f / 16777216.0f); /* units of pi */
          const int kn = (int)(((long long)k * nmod) % c->fft_len);
          const float arg = 2.0f * (float)kn / (float)c->fft_len + phase2;     /* units of pi, fp32 as on device */
          re = fmaf(amp, (float)cos(M_PI * (double)arg), re);
          im = fmaf(amp, (float)sin(M_PI * (double)arg), im);
        }
      }
      if (sc->adc_bits > 0) { /* the radio's integer samples (crn_synth_cfg.adc_bits) */
        const float scale = (float)(1 << (sc->adc_bits - 1));
        re = fminf(fmaxf(rintf(re * scale), -scale), scale - 1.f) / scale;
        im = fminf(fmaxf(rintf(im * scale), -scale), scale - 1.f) / scale;
      }
      iq[2 * i] = re;
      iq[2 * i + 1] = im;
    }
  }
  free(begin);
  free(bins);
  free(c2);
  return 0;
}
