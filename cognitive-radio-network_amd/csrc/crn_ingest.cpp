// crn_ingest.cpp — packet ingest ring (include/crn_sense.h, "ingest ring").
//
// Host-side counterpart of the ECR's rx -> CE hand-off
// (reference: src/extensible_cognitive_radio.cpp:1310-1324): per-stream staging of K packets,
// complete epochs packed into one of two pinned batch buffers, asynchronous H2D + sensing kernel +
// D2H on a private HIP stream, completion detected with an event so the caller never blocks.
#include <hip/hip_runtime.h>

#include <cstring>
#include <deque>
#include <string>
#include <vector>

#include "../../include/crn_sense.h"
#include "crn_internal.h"

#define HIP_TRY(expr)                                                                      \
  do {                                                                                     \
    hipError_t _e = (expr);                                                                \
    if (_e != hipSuccess)                                                                  \
      return crn::fail(CRN_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(_e)); \
  } while (0)

namespace {

struct SlotTag {
  int32_t stream;
  int64_t seq;
};

struct Batch {
  float *h_iq = nullptr;        // pinned [B][K][L] interleaved
  float *d_iq = nullptr;
  char *h_res = nullptr;        // pinned results: features | ann | decision | occupancy
  char *d_res = nullptr;
  hipEvent_t done = nullptr;
  std::vector<SlotTag> tags;    // slot -> (stream, seq) of the batch as launched / being filled
  int filled = 0;               // complete epochs staged
  bool in_flight = false;
};

}  // namespace

struct crn_ingest {
  crn_handle *h = nullptr;
  crn_cfg cfg;
  int n_streams = 0, L = 0, B = 0, K = 0;
  size_t epoch_floats = 0;                 // K * L * 2
  size_t off_ann = 0, off_dec = 0, off_occ = 0, res_bytes = 0;
  hipStream_t stream = nullptr;
  Batch batch[2];
  int fill = 0;                            // batch being filled
  std::vector<std::vector<float>> staging; // per stream: K packets of the running epoch
  std::vector<int> packets;                // per stream: packets staged
  std::vector<int64_t> seq;                // per stream: epochs completed
  std::deque<crn_epoch_result> ready;
};

// defined in crn_api.cpp
extern "C" int crn_sense_cfg_of(crn_handle *h, crn_cfg *out);

namespace {

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Move the results of a finished batch into the ready queue.
void harvest(crn_ingest *g, Batch &b) {
  const int nb = g->cfg.n_bands;
  const float *feat = reinterpret_cast<const float *>(b.h_res);
  const double *ann = reinterpret_cast<const double *>(b.h_res + g->off_ann);
  const int32_t *dec = reinterpret_cast<const int32_t *>(b.h_res + g->off_dec);
  const uint8_t *occ = reinterpret_cast<const uint8_t *>(b.h_res + g->off_occ);
  for (int s = 0; s < b.filled; s++) {
    crn_epoch_result r;
    std::memset(&r, 0, sizeof(r));
    r.stream = b.tags[s].stream;
    r.epoch_seq = b.tags[s].seq;
    r.decision = dec[s];
    if (g->cfg.decide == CRN_DECIDE_ANN) std::memcpy(r.ann_out, ann + 3 * s, 3 * sizeof(double));
    std::memcpy(r.features, feat + (size_t)s * nb, nb * sizeof(float));
    std::memcpy(r.occupancy, occ + (size_t)s * nb, nb);
    g->ready.push_back(r);
  }
  b.filled = 0;
  b.in_flight = false;
}

int launch(crn_ingest *g) {
  Batch &b = g->batch[g->fill];
  if (b.filled == 0) return CRN_OK;
  const int n = b.filled;
  HIP_TRY(hipMemcpyAsync(b.d_iq, b.h_iq, (size_t)n * g->epoch_floats * sizeof(float), hipMemcpyHostToDevice, g->stream));
  crn_out o;
  o.features = reinterpret_cast<float *>(b.d_res);
  o.ann_out = reinterpret_cast<double *>(b.d_res + g->off_ann);
  o.decision = reinterpret_cast<int32_t *>(b.d_res + g->off_dec);
  o.occupancy = reinterpret_cast<uint8_t *>(b.d_res + g->off_occ);
  o.spectrum = nullptr;
  if (int rc = crn_sense_run_device(g->h, b.d_iq, n, g->L, 0, &o, g->stream)) return rc;
  HIP_TRY(hipMemcpyAsync(b.h_res, b.d_res, g->res_bytes, hipMemcpyDeviceToHost, g->stream));
  HIP_TRY(hipEventRecord(b.done, g->stream));
  b.in_flight = true;
  // switch to the other buffer; if it is still in flight, wait for it (back-pressure)
  g->fill ^= 1;
  Batch &nb = g->batch[g->fill];
  if (nb.in_flight) {
    HIP_TRY(hipEventSynchronize(nb.done));
    harvest(g, nb);
  }
  return CRN_OK;
}

}  // namespace

extern "C" {

int crn_ingest_create(crn_handle *h, int32_t n_streams, int32_t samples_per_packet, int32_t epochs_per_batch,
                      crn_ingest **out) {
  if (!h || !out) return crn::fail(CRN_ERR_ARG, "crn_ingest_create: null argument");
  *out = nullptr;
  crn_cfg cfg;
  if (int rc = crn_sense_cfg_of(h, &cfg)) return rc;
  if (n_streams < 1 || epochs_per_batch < 1) return crn::fail(CRN_ERR_ARG, "n_streams / epochs_per_batch < 1");
  if (cfg.hop != cfg.fft_len) return crn::fail(CRN_ERR_ARG, "the ingest ring takes disjoint frames (hop == fft_len)");
  if (samples_per_packet < 1 || samples_per_packet > cfg.fft_len)
    return crn::fail(CRN_ERR_ARG, "samples_per_packet must be in 1..fft_len");
  HIP_TRY(hipSetDevice(cfg.device));
  crn_ingest *g = new (std::nothrow) crn_ingest();
  if (!g) return crn::fail(CRN_ERR_NOMEM, "out of host memory");
  g->h = h;
  g->cfg = cfg;
  g->n_streams = n_streams;
  g->L = samples_per_packet;
  g->B = epochs_per_batch;
  g->K = cfg.frames_per_epoch;
  g->epoch_floats = (size_t)g->K * g->L * 2;
  const size_t nb = (size_t)cfg.n_bands;
  g->off_ann = align_up((size_t)g->B * nb * sizeof(float), 256);
  g->off_dec = g->off_ann + align_up((size_t)g->B * 3 * sizeof(double), 256);
  g->off_occ = g->off_dec + align_up((size_t)g->B * sizeof(int32_t), 256);
  g->res_bytes = g->off_occ + align_up((size_t)g->B * nb, 256);
  g->staging.assign(n_streams, std::vector<float>(g->epoch_floats, 0.f));
  g->packets.assign(n_streams, 0);
  g->seq.assign(n_streams, 0);
  hipError_t e = hipStreamCreateWithFlags(&g->stream, hipStreamNonBlocking);
  for (int i = 0; i < 2 && e == hipSuccess; i++) {
    Batch &b = g->batch[i];
    b.tags.resize(g->B);
    const size_t iq_bytes = (size_t)g->B * g->epoch_floats * sizeof(float);
    if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void **>(&b.h_iq), iq_bytes, hipHostMallocDefault);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&b.d_iq), iq_bytes);
    if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void **>(&b.h_res), g->res_bytes, hipHostMallocDefault);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&b.d_res), g->res_bytes);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&b.done, hipEventDisableTiming);
  }
  if (e != hipSuccess) {
    crn_ingest_destroy(g);
    return crn::fail(CRN_ERR_NOMEM, std::string("crn_ingest_create: ") + hipGetErrorString(e));
  }
  *out = g;
  return CRN_OK;
}

int crn_ingest_push(crn_ingest *g, int32_t stream, const float *iq_packet) {
  if (!g || !iq_packet) return crn::fail(CRN_ERR_ARG, "crn_ingest_push: null argument");
  if (stream < 0 || stream >= g->n_streams) return crn::fail(CRN_ERR_ARG, "stream id out of range");
  std::vector<float> &st = g->staging[stream];
  std::memcpy(st.data() + (size_t)g->packets[stream] * g->L * 2, iq_packet, (size_t)g->L * 2 * sizeof(float));
  if (++g->packets[stream] < g->K) return CRN_OK;
  // epoch complete: pack it into the batch being filled
  g->packets[stream] = 0;
  Batch &b = g->batch[g->fill];
  std::memcpy(b.h_iq + (size_t)b.filled * g->epoch_floats, st.data(), g->epoch_floats * sizeof(float));
  b.tags[b.filled] = SlotTag{stream, g->seq[stream]++};
  if (++b.filled == g->B) return launch(g);
  return CRN_OK;
}

int crn_ingest_flush(crn_ingest *g) {
  if (!g) return crn::fail(CRN_ERR_ARG, "null ingest ring");
  return launch(g);
}

int crn_ingest_poll(crn_ingest *g, crn_epoch_result *out, int32_t max_results, int32_t *n_out) {
  if (!g || !n_out || (max_results > 0 && !out)) return crn::fail(CRN_ERR_ARG, "crn_ingest_poll: null argument");
  // the batch launched before the current fill buffer's twin is the older one
  for (int k = 0; k < 2; k++) {
    Batch &b = g->batch[g->fill ^ 1 ^ k];
    if (!b.in_flight) continue;
    hipError_t q = hipEventQuery(b.done);
    if (q == hipSuccess) harvest(g, b);
    else if (q != hipErrorNotReady) return crn::fail(CRN_ERR_DEVICE, std::string("hipEventQuery: ") + hipGetErrorString(q));
  }
  int n = 0;
  while (n < max_results && !g->ready.empty()) {
    out[n++] = g->ready.front();
    g->ready.pop_front();
  }
  *n_out = n;
  return CRN_OK;
}

int crn_ingest_drain(crn_ingest *g) {
  if (!g) return crn::fail(CRN_ERR_ARG, "null ingest ring");
  if (int rc = launch(g)) return rc;
  HIP_TRY(hipStreamSynchronize(g->stream));
  // harvest in launch order: the buffer that is NOT the fill buffer was launched last
  Batch &older = g->batch[g->fill];
  Batch &newer = g->batch[g->fill ^ 1];
  if (older.in_flight) harvest(g, older);
  if (newer.in_flight) harvest(g, newer);
  return CRN_OK;
}

int crn_ingest_destroy(crn_ingest *g) {
  if (!g) return CRN_OK;
  if (g->stream) (void)hipStreamSynchronize(g->stream);
  for (int i = 0; i < 2; i++) {
    Batch &b = g->batch[i];
    if (b.h_iq) (void)hipHostFree(b.h_iq);
    if (b.d_iq) (void)hipFree(b.d_iq);
    if (b.h_res) (void)hipHostFree(b.h_res);
    if (b.d_res) (void)hipFree(b.d_res);
    if (b.done) (void)hipEventDestroy(b.done);
  }
  if (g->stream) (void)hipStreamDestroy(g->stream);
  delete g;
  return CRN_OK;
}

}  // extern "C"
