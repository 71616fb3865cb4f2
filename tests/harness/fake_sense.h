// fake_sense.h — TEST INFRASTRUCTURE: what csrc/crn_ingest.cpp (and the engine) link against in libcrnsense, restated for
// CPU-only unit tests (ring_unit.cpp, engine_unit.cpp) together with tests/harness/fake_hip.  The sensing stand-in:
//   features[0] of an epoch = sum of all its samples, features[1] = its first sample, decision = L (ring_unit) or
//   the rounded first sample (engine_unit: g_fake_decision_from_data), ann_out = {0, 1, 2}; with threshold plans the "decision"
//   d also marks occupancy[d] (4-band plans) or every band of channel d's first reference bin (64-band plans), so that the
//   engine's channel mapping can be watched.  Overlapped frames (hop < fft_len): an epoch is the contiguous run of
//   (K - 1) hop + N samples starting `epoch_stride` samples after the previous one, L must be fft_len.
#ifndef CRN_FAKE_SENSE_H
#define CRN_FAKE_SENSE_H
#include <string.h>

#include <atomic>
#include <string>

#include <hip/hip_runtime.h>   // the stand-in under tests/harness/fake_hip

#include "../../include/crn_sense.h"
#include "../../include/crn_sense_sc16.h"   // (the stand-in library carries the optional wire-format entry points too)

std::atomic<long long> g_fake_gpu_latency_ns(0);

// ---- what crn_ingest.cpp links against in libcrnsense, restated for the test -----------------------------------------
namespace crn {
static thread_local std::string g_err;
int fail(int code, const std::string &msg) { g_err = msg; return code; }
}  // namespace crn
static std::atomic<int> g_fail_next_launch{0};
static std::atomic<int> g_fake_decision_from_data{0};
extern "C" {
const char *crn_last_error(void) { return crn::g_err.c_str(); }
struct crn_handle { crn_cfg cfg; };
int crn_sense_cfg_of(crn_handle *h, crn_cfg *out) { *out = h->cfg; return CRN_OK; }
static std::atomic<int> g_fake_rings_attached{0};
int crn_sense_ring_count(crn_handle *, int delta) { g_fake_rings_attached += delta; return CRN_OK; }
static std::atomic<long> g_fake_warm_launches{0};
int crn_sense_warm_stream(crn_handle *, void *) { g_fake_warm_launches++; return CRN_OK; }
// calibration (crn_ingest_calibrate -> the ring's launcher thread -> these two): "the median band energy" of the stand-in's features
// is the mean of features[1] (the epochs' first samples: what the test put there).  The stand-in makes the kinds of HIP call the
// real one makes — an upload, a wait — through the fake runtime, so a test that watches a thread sees them if they run on it.
static float g_fake_thresholds_set[CRN_MAX_BANDS];
static std::atomic<int> g_fake_threshold_updates{0}, g_fake_nf_reserved{0};
int crn_sense_reserve_noise_floor(crn_handle *) { g_fake_nf_reserved++; return CRN_OK; }
int crn_sense_calibrate_thresholds(crn_handle *h, const float *features, int64_t n_epochs, float lambda, float *nf_out, void *stream) {
  if (!g_fake_nf_reserved.load()) return crn::fail(CRN_ERR_STATE, "crn_sense_calibrate_thresholds without crn_sense_reserve_noise_floor");
  static float upload[4096 * CRN_MAX_BANDS];
  (void)hipMemcpyAsync(upload, features, sizeof(float) * (size_t)n_epochs * h->cfg.n_bands, hipMemcpyHostToDevice, static_cast<hipStream_t>(stream));
  (void)hipStreamSynchronize(static_cast<hipStream_t>(stream));
  double s = 0;
  for (int64_t e = 0; e < n_epochs; e++) s += upload[e * h->cfg.n_bands + 1];
  *nf_out = (float)(s / (double)n_epochs);
  for (int b = 0; b < h->cfg.n_bands; b++) g_fake_thresholds_set[b] = h->cfg.thresh[b] = lambda * *nf_out;
  g_fake_threshold_updates++;
  return CRN_OK;
}
static std::atomic<long long> g_fake_launches{0};
static std::atomic<int> g_fake_last_L{0};
static std::atomic<long long> g_fake_last_stride{0};
int crn_sense_run_device(crn_handle *h, const float *d_iq, int64_t n_epochs, int32_t L, int64_t stride, const crn_out *o, void *) {
  if (g_fail_next_launch.exchange(0)) return crn::fail(CRN_ERR_DEVICE, "forced launch failure");
  const int K = h->cfg.frames_per_epoch, nb = h->cfg.n_bands;
  const bool overlapped = h->cfg.hop != h->cfg.fft_len;
  if (overlapped && L != h->cfg.fft_len) return crn::fail(CRN_ERR_ARG, "overlapped frames need samples_per_frame == fft_len");
  const long long span = overlapped ? (long long)(K - 1) * h->cfg.hop + h->cfg.fft_len : (long long)K * L;
  if (stride <= 0) stride = overlapped ? (long long)K * h->cfg.hop : (long long)K * L;
  g_fake_launches++;
  g_fake_last_L = L;
  g_fake_last_stride = stride;
  for (int64_t e = 0; e < n_epochs; e++) {
    const float *x = d_iq + (size_t)e * stride * 2;
    double s = 0;
    for (long long i = 0; i < span * 2; i++) s += x[i];
    if (o->features) {   // any output may be NULL (crn_out)
      for (int b = 0; b < nb; b++) o->features[e * nb + b] = 0.f;
      o->features[e * nb + 0] = (float)s;
      o->features[e * nb + 1] = x[0];
    }
    const int d = g_fake_decision_from_data.load() ? (int)(x[0] + 0.5f) : L;
    if (o->decision) o->decision[e] = d;
    if (o->ann_out)
      for (int k = 0; k < 3; k++) o->ann_out[e * 3 + k] = (double)k;
    if (o->occupancy) {
      memset(o->occupancy + e * nb, 0, (size_t)nb);
      if (g_fake_decision_from_data.load() && h->cfg.decide == CRN_DECIDE_THRESHOLD && d >= 1 && d <= 3) {
        // 4-band plans {NF, CH1, CH2, CH3}: band d; equal-band plans: the band holding the first bin of the reference's channel d
        // (CE_Predictive_Node.cpp:173-191: CH1 from bin 0, CH2 from 55, CH3 from 189 of 512)
        const int first_bin[4] = {0, 0, 55, 189};
        const int b = nb == 4 ? d : first_bin[d] * (h->cfg.fft_len / 512) / (h->cfg.fft_len / nb);
        o->occupancy[e * nb + b] = 1;
      }
    }
  }
  return CRN_OK;
}
// wire-format twin of the stand-in: the same "features" from int16 pairs
int crn_sense_run_device_sc16(crn_handle *h, const int16_t *d_iq, int64_t n_epochs, int32_t L, int64_t, const crn_out *o, void *) {
  if (g_fail_next_launch.exchange(0)) return crn::fail(CRN_ERR_DEVICE, "forced launch failure");
  const int K = h->cfg.frames_per_epoch, nb = h->cfg.n_bands;
  for (int64_t e = 0; e < n_epochs; e++) {
    const int16_t *x = d_iq + (size_t)e * K * L * 2;
    double s = 0;
    for (int i = 0; i < K * L * 2; i++) s += x[i];
    if (o->features) {
      for (int b = 0; b < nb; b++) o->features[e * nb + b] = 0.f;
      o->features[e * nb + 0] = (float)s;
      o->features[e * nb + 1] = (float)x[0];
    }
    if (o->decision) o->decision[e] = -L;   // negative: the wire-format launch ran
    if (o->ann_out)
      for (int k = 0; k < 3; k++) o->ann_out[e * 3 + k] = (double)k;
    if (o->occupancy) memset(o->occupancy + e * nb, 0, (size_t)nb);
  }
  return CRN_OK;
}
// what the ring calls (csrc/crn_ingest.cpp): either sample format
int crn_sense_run_device_any(crn_handle *h, const void *d_iq, int32_t bytes_per_sample, int64_t n_epochs, int32_t L, int64_t stride, const crn_out *o, void *st) {
  return bytes_per_sample == 4 ? crn_sense_run_device_sc16(h, static_cast<const int16_t *>(d_iq), n_epochs, L, stride, o, st)
                               : crn_sense_run_device(h, static_cast<const float *>(d_iq), n_epochs, L, stride, o, st);
}
}

#endif
