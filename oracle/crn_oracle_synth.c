/* crn_oracle_synth.c — TEST INFRASTRUCTURE ONLY (see crn_oracle.c's header): CPU twin of the device
 * signal generator (crn_synth_fill_device_ex, csrc/crn_kernels.hip synth_kernel / pu_pattern_kernel).
 *
 * The generator has no counterpart in the reference to be exact against — it is this build's
 * measurement aid, shaped after the reference's traffic sources:
 *   uniform channel pick ......... cognitive_engines/CE_Random_Behaviour_PU/CE_Random_Behaviour_PU.cpp:41-53
 *   Markov chain, as written ..... cognitive_engines/CE_PU_MARKOV_Chain_Tx/CE_PU_MARKOV_Chain_Tx.cpp:82-128
 *   CW / noise / multicarrier .... src/interferer.cpp:128-140, 248-282
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

  /* occupancy pattern */
  if (sc->pu_model != CRN_PU_UNIFORM) {
    if (n_epochs % sc->n_streams != 0 || n_active < 1) { free(begin); free(bins); return -1; }
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
      if (pick > 0) {
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
      iq[2 * i] = re;
      iq[2 * i + 1] = im;
    }
  }
  free(begin);
  free(bins);
  return 0;
}
