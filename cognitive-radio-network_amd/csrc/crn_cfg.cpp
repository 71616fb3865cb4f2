// crn_cfg.cpp — configuration helpers of libcrnsense (host only, no device code).
//
// The reference hard-codes every sensing parameter (CE_Predictive_Node.hpp:30-33,42-43,55-57;
// CE_Predictive_Node.cpp:78-120,173-191,245-261).  crn_cfg_reference() restates them as data so
// the same kernel serves the reference-exact mode and the BASELINE.json generalisations.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <string>

#include "../../include/crn_sense.h"
#include "crn_internal.h"

namespace {

// (lo, hi, band) in the order the reference sums them (CE_Predictive_Node.cpp:173-191).
// band ids follow Features_Buffer[1..4] = {NOISE_FLOOR, CH1, CH2, CH3} (.cpp:200).
struct RefSeg { int lo, hi, band; };
constexpr RefSeg kRefSegs[5] = {
    {0, 16, 1},     // .cpp:173  M1 += fft_avg[0..15]
    {496, 511, 1},  // .cpp:177  M1 += fft_avg[496..510]   (bin 511 is not summed)
    {55, 85, 2},    // .cpp:181  M2
    {189, 222, 3},  // .cpp:185  M3
    {300, 310, 0},  // .cpp:189  NF
};

// WeightIH[i][j] (.cpp:78-102) and WeightHO[j][k] (.cpp:103-120); row/column 0 of the hidden /
// output index is never read by the reference and is left 0 here.
constexpr double kWih[CRN_ANN_IN + 1][CRN_ANN_HID + 1] = {
    {0.0, -0.188208, -0.170684, -0.024726, 0.001448, 0.015983},
    {0.0, -0.106634, -0.415470, 0.309261, 0.159974, 0.212781},
    {0.0, 0.005650, 0.741944, 0.006133, -0.620100, 0.669892},
    {0.0, -0.057578, 0.621154, -0.048268, -0.249186, 0.734475},
    {0.0, 0.092680, 0.809336, -0.010821, -0.546496, 0.609384},
};
constexpr double kWho[CRN_ANN_HID + 1][CRN_ANN_OUT + 1] = {
    {0.0, -7.033320, 2.726400, -2.590206},  {0.0, 10.857465, -18.452471, 15.609466},
    {0.0, -6.848443, 2.053071, -2.929559},  {0.0, 17.053079, -13.375309, -15.703407},
    {0.0, 0.087664, -0.269499, 0.407028},   {0.0, -6.552455, 2.655529, -2.552555},
};

void fill_common(crn_cfg *c) {
  std::memset(c, 0, sizeof(*c));
  c->abi_version = CRN_ABI_VERSION;
  c->ref_band = -1;
  std::memcpy(c->ann_w_ih, kWih, sizeof(kWih));
  std::memcpy(c->ann_w_ho, kWho, sizeof(kWho));
  c->ann_threshold = 0.8;              // .cpp:245,250,255
  c->tx_freq_for_decision[0] = 0.0;    // "ALL BUSY": no set_tx_freq call (.cpp:260-261)
  c->tx_freq_for_decision[1] = 835e6;  // CH1 occupied -> CHANNEL2 (.cpp:247, .hpp:56)
  c->tx_freq_for_decision[2] = 833e6;  // CH2 occupied -> CHANNEL1 (.cpp:252, .hpp:55)
  c->tx_freq_for_decision[3] = 835e6;  // CH3 occupied -> CHANNEL2 (.cpp:257)
}

}  // namespace

extern "C" {

int crn_cfg_reference(crn_cfg *cfg) {
  if (!cfg) return crn::fail(CRN_ERR_ARG, "crn_cfg_reference: null cfg");
  fill_common(cfg);
  cfg->fft_len = 512;           // fft_length     .hpp:31
  cfg->frames_per_epoch = 10;   // fft_averaging  .hpp:32
  cfg->hop = 512;
  cfg->mode = CRN_MODE_REF_MAG;
  cfg->decide = CRN_DECIDE_ANN;
  cfg->window = CRN_WINDOW_RECT;
  cfg->n_bands = 4;
  cfg->n_segs = 5;
  for (int i = 0; i < 5; i++) cfg->segs[i] = crn_band_seg{kRefSegs[i].lo, kRefSegs[i].hi, kRefSegs[i].band};
  return CRN_OK;
}

int crn_cfg_reference_scaled(crn_cfg *cfg, int32_t fft_len) {
  if (int rc = crn_cfg_reference(cfg)) return rc;
  if (fft_len < 512 || fft_len % 512 != 0) return crn::fail(CRN_ERR_ARG, "fft_len must be a multiple of 512");
  const int s = fft_len / 512;
  cfg->fft_len = fft_len;
  cfg->hop = fft_len;
  for (int i = 0; i < 5; i++) cfg->segs[i] = crn_band_seg{kRefSegs[i].lo * s, kRefSegs[i].hi * s, kRefSegs[i].band};
  return CRN_OK;
}

// Weights file: plain text, '#' starts a comment, then 30 + 24 numbers in the reference's array order —
// WeightIH[i][j], i = 0..4, j = 0..5 (row-major, column 0 unused) and WeightHO[j][k], j = 0..5, k = 0..3 (column 0 unused) —
// optionally followed by the decision threshold.  %.17g round-trips a double exactly.
int crn_cfg_save_ann(const crn_cfg *cfg, const char *path) {
  if (!cfg || !path) return crn::fail(CRN_ERR_ARG, "crn_cfg_save_ann: null argument");
  FILE *f = std::fopen(path, "w");
  if (!f) return crn::fail(CRN_ERR_ARG, std::string("crn_cfg_save_ann: cannot write ") + path);
  std::fprintf(f, "# libcrnsense 4-5-3 network: WeightIH[5][6] then WeightHO[6][4] (reference indexing, CE_Predictive_Node.hpp:66,71), then the threshold\n");
  for (int i = 0; i <= CRN_ANN_IN; i++) {
    for (int j = 0; j <= CRN_ANN_HID; j++) std::fprintf(f, "%.17g ", cfg->ann_w_ih[i][j]);
    std::fprintf(f, "\n");
  }
  for (int j = 0; j <= CRN_ANN_HID; j++) {
    for (int k = 0; k <= CRN_ANN_OUT; k++) std::fprintf(f, "%.17g ", cfg->ann_w_ho[j][k]);
    std::fprintf(f, "\n");
  }
  std::fprintf(f, "%.17g\n", cfg->ann_threshold);
  const bool ok = std::ferror(f) == 0;
  if (std::fclose(f) != 0 || !ok) return crn::fail(CRN_ERR_ARG, std::string("crn_cfg_save_ann: write error on ") + path);
  return CRN_OK;
}

int crn_cfg_load_ann(crn_cfg *cfg, const char *path) {
  if (!cfg || !path) return crn::fail(CRN_ERR_ARG, "crn_cfg_load_ann: null argument");
  FILE *f = std::fopen(path, "r");
  if (!f) return crn::fail(CRN_ERR_ARG, std::string("crn_cfg_load_ann: cannot read ") + path);
  double v[56];
  int n = 0;
  char tok[128];
  while (n < 56 && std::fscanf(f, "%127s", tok) == 1) {   // (a 56th number is one too many: counted so that it is refused)
    if (tok[0] == '#') {  // comment: skip the rest of the line
      int ch;
      while ((ch = std::fgetc(f)) != EOF && ch != '\n') {}
      continue;
    }
    char *end = nullptr;
    const double x = std::strtod(tok, &end);
    if (end == tok || *end != '\0' || !std::isfinite(x)) {
      std::fclose(f);
      return crn::fail(CRN_ERR_ARG, std::string("crn_cfg_load_ann: not a finite number: '") + tok + "' in " + path);
    }
    v[n++] = x;
  }
  std::fclose(f);
  if (n != 54 && n != 55)
    return crn::fail(CRN_ERR_ARG, std::string("crn_cfg_load_ann: expected 54 weights (+ threshold), found ") + std::to_string(n) + " numbers in " + path);
  std::memcpy(cfg->ann_w_ih, v, sizeof(cfg->ann_w_ih));
  std::memcpy(cfg->ann_w_ho, v + 30, sizeof(cfg->ann_w_ho));
  if (n == 55) cfg->ann_threshold = v[54];
  return CRN_OK;
}

int crn_cfg_energy_scaled(crn_cfg *cfg, int32_t fft_len, float lambda) {
  if (!cfg) return crn::fail(CRN_ERR_ARG, "crn_cfg_energy_scaled: null cfg");
  if (fft_len < 512 || fft_len % 512 != 0) return crn::fail(CRN_ERR_ARG, "fft_len must be a multiple of 512");
  fill_common(cfg);
  const int s = fft_len / 512;
  cfg->fft_len = fft_len;
  cfg->frames_per_epoch = 10;
  cfg->hop = fft_len;
  cfg->mode = CRN_MODE_ENERGY;
  cfg->decide = CRN_DECIDE_THRESHOLD;
  cfg->window = CRN_WINDOW_RECT;
  cfg->n_bands = 4;
  cfg->n_segs = 5;
  int bins[4] = {0, 0, 0, 0};
  for (int i = 0; i < 5; i++) {
    cfg->segs[i] = crn_band_seg{kRefSegs[i].lo * s, kRefSegs[i].hi * s, kRefSegs[i].band};
    bins[kRefSegs[i].band] += (kRefSegs[i].hi - kRefSegs[i].lo) * s;
  }
  cfg->ref_band = 0;
  cfg->thresh[0] = std::numeric_limits<float>::infinity();  // the noise-floor band is never "occupied"
  for (int b = 1; b < 4; b++) cfg->thresh[b] = lambda * (float)bins[b] / (float)bins[0];
  return CRN_OK;
}

int crn_cfg_welch(crn_cfg *cfg, int32_t fft_len, int32_t frames_per_epoch, int32_t n_bands) {
  if (!cfg) return crn::fail(CRN_ERR_ARG, "crn_cfg_welch: null cfg");
  if (fft_len < 512 || n_bands < 1 || n_bands > CRN_MAX_BANDS || fft_len % n_bands != 0 ||
      frames_per_epoch < 1)
    return crn::fail(CRN_ERR_ARG, "crn_cfg_welch: bad fft_len / n_bands / frames_per_epoch");
  fill_common(cfg);
  cfg->fft_len = fft_len;
  cfg->frames_per_epoch = frames_per_epoch;
  cfg->hop = fft_len / 2;
  cfg->mode = CRN_MODE_ENERGY;
  cfg->decide = CRN_DECIDE_THRESHOLD;
  cfg->window = CRN_WINDOW_HANN;
  cfg->n_bands = n_bands;
  cfg->n_segs = n_bands;
  const int w = fft_len / n_bands;
  for (int b = 0; b < n_bands; b++) {
    cfg->segs[b] = crn_band_seg{b * w, (b + 1) * w, b};
    cfg->thresh[b] = std::numeric_limits<float>::infinity();
  }
  cfg->ref_band = -1;
  return CRN_OK;
}

int crn_cfg_welch_scaled(crn_cfg *cfg, int32_t fft_len, int32_t frames_per_epoch, float lambda) {
  if (int rc = crn_cfg_energy_scaled(cfg, fft_len, lambda)) return rc;
  if (frames_per_epoch < 1) return crn::fail(CRN_ERR_ARG, "crn_cfg_welch_scaled: frames_per_epoch < 1");
  cfg->frames_per_epoch = frames_per_epoch;
  cfg->hop = fft_len / 2;
  cfg->window = CRN_WINDOW_HANN;
  return CRN_OK;
}

}  // extern "C"
