// crn_api.cpp — the C ABI of libcrnsense (include/crn_sense.h): handle, device tables, launches.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/crn_sense.h"
#ifdef CRN_WITH_SC16
#include "../../include/crn_sense_sc16.h"   // the optional wire-format entry points (libcrnsense_sc16.so)
#endif
#include "crn_internal.h"
#include "crn_kernels.h"

namespace crn {
static thread_local std::string g_err;
int fail(int code, const std::string &msg) {
  g_err = msg;
  return code;
}
}  // namespace crn

#define HIP_TRY(expr)                                                                          \
  do {                                                                                         \
    hipError_t _e = (expr);                                                                    \
    if (_e != hipSuccess)                                                                      \
      return crn::fail(CRN_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(_e));     \
  } while (0)

struct crn_handle {
  explicit crn_handle(int dev) : device(dev) {}
  // The HIP device, fixed at creation: what every entry point makes current, readable without a lock (cfg.device is the same number,
  // but cfg as a whole is rewritten under tables_mu by crn_sense_set_bands while an ingest ring's launcher thread may be in here).
  const int device;
  crn_cfg cfg;
  int variant = 0;
  int groups_per_wg = 0;        // 0 = automatic
  int64_t tail_groups = -1;     // < 0 = automatic; epoch groups handed to the short tail workgroups at the end
  int tail_groups_per_wg = 0;   // 0 = automatic; epoch groups per tail workgroup
  std::atomic<int64_t> n_dealt{0};   // launches that ran the dealt-frame kernel (crn_sense_dealt_launches)
  int64_t deal_max_epochs = -1; // launches of up to this many epochs run the dealt-frame kernel where it exists (< 0: automatic, from n_cus)
  int n_row_entries = 0;        // > 0: the band plan qualifies for register-resident band sums
  int aligned_shift = 0;        // N = 4096, equal contiguous bands of 64 / 128 / 256 bins in order: log2 of the width
  std::atomic<int> n_rings{0};  // ingest rings created on this handle (they size their result buffers for cfg.n_bands)
  int n_cus = 256;              // compute units of the device (workgroup slots = n_cus x workgroups per CU): read at creation
  size_t lds_budget = 64 * 1024;   // LDS a workgroup may ask for on this device (hipDeviceAttributeMaxSharedMemoryPerBlock: 160 KiB on gfx950)
  unsigned acc_mask = 0xFFFFu;  // accumulator registers (bit j R3 + d) that hold a bin of some band (N = 4096: the 256-bin rows)
  // one device slab holding every table
  void *d_tables = nullptr;
  const float2 *d_tw1 = nullptr, *d_tw2 = nullptr;
  const float *d_window = nullptr, *d_thresh = nullptr;
  const int *d_band_seg_begin = nullptr, *d_seg_lo = nullptr, *d_seg_hi = nullptr;
  const int *d_band_bins_begin = nullptr, *d_band_bins = nullptr, *d_band_tab = nullptr, *d_band_c2 = nullptr;
  const double *d_wih = nullptr, *d_who = nullptr;
  // scratch of crn_sense_run_host
  void *d_scratch = nullptr;
  size_t scratch_bytes = 0;
  double window_power = 0.0;   // sum of the squared fp32 window values (crn_monitor_rows_device)
  double wire_full_scale = 32768.0;   // crn_sense_set_wire_full_scale
  float *d_nf_scratch = nullptr;   // crn_noise_floor_device: per-epoch medians + the result
  void *h_small = nullptr;     // pinned in-place buffer of run_host's small batches (samples | results)
  size_t h_small_bytes = 0;
  void *h_results = nullptr;   // pinned staging for the per-epoch results of run_host (one D2H)
  size_t h_results_bytes = 0;
  // Live updates against launches from other threads (an ingest ring's launcher thread calls run_device_impl while the thread that
  // owns the handle calls crn_sense_set_bands / _set_thresholds / _set_ann): `tables_mu` covers cfg, every table pointer and the
  // plan-derived fields above.  A launch holds it from the first read of cfg until the kernel is enqueued, an update from its
  // first write until its copies are enqueued (set_bands: until the old slab is freed) — so a launch sees one plan, whole, and no
  // launch can pick up a slab after the update that frees it has started.
  std::mutex tables_mu;
  // The noise-floor scratch and upload buffers (d_nf_scratch, h_nf_features, d_nf_features) and the blocking reductions that use them:
  // a lock of their own, so that a calibration waiting for the device never holds tables_mu — launches on other threads go on.
  // Order: nf_mu before tables_mu.
  std::mutex nf_mu;
  // Pinned staging of the small asynchronous updates (thresholds, weights): hipMemcpyAsync reads its source when the stream gets
  // there, so each update copies from a slot of its own that is not rewritten until the event behind its copies has completed.
  struct UpdateSlot {
    float thresh[CRN_MAX_BANDS];
    double w_ih[CRN_ANN_IN + 1][CRN_ANN_HID + 1];
    double w_ho[CRN_ANN_HID + 1][CRN_ANN_OUT + 1];
  };
  static constexpr int kUpdateSlots = 8;
  UpdateSlot *upd = nullptr;                 // pinned [kUpdateSlots]
  hipEvent_t upd_done[kUpdateSlots] = {};
  bool upd_used[kUpdateSlots] = {};
  int64_t upd_next = 0;
  float *h_nf_features = nullptr;            // pinned upload buffer of crn_noise_floor_host (crn_sense_reserve_noise_floor)
  float *d_nf_features = nullptr;
  // counters (crn_sense_get_stats): launches come from the caller's thread or from an ingest ring's launcher thread
  std::atomic<int64_t> n_launches{0}, n_epochs{0}, n_samples{0};
  std::mutex timing_mu;        // everything below
  bool timing = false;
  static constexpr int kTimedSlots = 16;
  hipEvent_t t_start[kTimedSlots] = {}, t_stop[kTimedSlots] = {};
  int64_t t_issued = 0, t_collected = 0, t_dropped = 0;   // slots [t_collected, t_issued) are in flight (mod kTimedSlots)
  double kernel_ms = 0.0, kernel_ms_last = 0.0, kernel_ms_min = 0.0, kernel_ms_max = 0.0;
};

namespace {

constexpr size_t kInPlaceBytes = 512 * 1024;   // run_host batches up to this size are read and written in place by the kernel

bool supported_n(int n) { return n == 512 || n == 1024 || n == 2048 || n == 4096; }

int validate(const crn_cfg *c) {
  if (!c) return crn::fail(CRN_ERR_ARG, "null cfg");
  if (c->abi_version != CRN_ABI_VERSION) return crn::fail(CRN_ERR_ARG, "cfg.abi_version mismatch");
  if (!supported_n(c->fft_len)) return crn::fail(CRN_ERR_ARG, "fft_len must be 512, 1024, 2048 or 4096");
  if (c->frames_per_epoch < 1) return crn::fail(CRN_ERR_ARG, "frames_per_epoch < 1");
  if (c->hop < 1 || c->hop > c->fft_len) return crn::fail(CRN_ERR_ARG, "hop out of range");
  if (c->mode != CRN_MODE_REF_MAG && c->mode != CRN_MODE_ENERGY) return crn::fail(CRN_ERR_ARG, "bad mode");
  if (c->decide < CRN_DECIDE_ANN || c->decide > CRN_DECIDE_NONE) return crn::fail(CRN_ERR_ARG, "bad decide");
  if (c->window < CRN_WINDOW_RECT || c->window > CRN_WINDOW_BLACKMAN_HARRIS) return crn::fail(CRN_ERR_ARG, "bad window");
  if (c->n_bands < 1 || c->n_bands > CRN_MAX_BANDS) return crn::fail(CRN_ERR_ARG, "n_bands out of range");
  if (c->n_segs < 1 || c->n_segs > CRN_MAX_SEGS) return crn::fail(CRN_ERR_ARG, "n_segs out of range");
  for (int s = 0; s < c->n_segs; s++) {
    const crn_band_seg &g = c->segs[s];
    if (g.lo < 0 || g.hi > c->fft_len || g.lo > g.hi || g.band < 0 || g.band >= c->n_bands)
      return crn::fail(CRN_ERR_ARG, "band segment " + std::to_string(s) + " out of range");
  }
  if (c->decide == CRN_DECIDE_ANN && c->n_bands != 4)
    return crn::fail(CRN_ERR_ARG, "DECIDE_ANN needs exactly 4 bands {NF, CH1, CH2, CH3}");
  if (c->decide == CRN_DECIDE_THRESHOLD && c->ref_band >= c->n_bands)
    return crn::fail(CRN_ERR_ARG, "ref_band out of range");
  return CRN_OK;
}

// exp(-2 pi j q / n) with the angle index reduced exactly and the trig done in double.
float2 twiddle(long long q, int n) {
  q %= n;
  const double ang = -2.0 * M_PI * (double)q / (double)n;
  return make_float2((float)std::cos(ang), (float)std::sin(ang));
}

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

}  // namespace

// Everything a handle keeps in HBM for its configuration, built from h->cfg into one fresh slab (crn_sense_create, and again by
// crn_sense_set_bands on a live handle): twiddles, window, the band plan in its three forms, thresholds, ANN weights.
static int build_tables(crn_handle *h) {
  const crn_cfg &cfg = h->cfg;
  // what this function derives from cfg besides the device slab: put back if the slab cannot be made (a live handle keeps working
  // with its old plan: crn_sense_set_bands)
  const double keep_wp = h->window_power;
  const int keep_as = h->aligned_shift, keep_nre = h->n_row_entries;
  const unsigned keep_am = h->acc_mask;
  auto undo = [&] { h->window_power = keep_wp; h->aligned_shift = keep_as; h->n_row_entries = keep_nre; h->acc_mask = keep_am; };
  h->window_power = 0.0;
  h->aligned_shift = 0;
  h->acc_mask = 0xFFFFu;
  const int N = cfg.fft_len, R3 = N / 256, T = N / 16;
  std::vector<float2> tw1((size_t)17 * T), tw2((size_t)16 * R3);  // row 16 of tw1: W_N^{16 t}
  for (int i = 0; i < 16; i++)
    for (int t = 0; t < T; t++) tw1[(size_t)i * T + t] = twiddle((long long)i * t, N);
  for (int t = 0; t < T; t++) tw1[(size_t)16 * T + t] = twiddle(16LL * t, N);
  for (int i = 0; i < 16; i++)
    for (int m = 0; m < R3; m++) tw2[(size_t)i * R3 + m] = twiddle((long long)i * m, T);
  std::vector<float> win(N, 1.0f);
  if (cfg.window == CRN_WINDOW_HANN)
    for (int n = 0; n < N; n++) win[n] = (float)(0.5 - 0.5 * std::cos(2.0 * M_PI * (double)n / (double)N));
  if (cfg.window == CRN_WINDOW_BLACKMAN_HARRIS)
    for (int n = 0; n < N; n++) {
      const double x = 2.0 * M_PI * (double)n / (double)(N - 1);
      win[n] = (float)(0.35875 - 0.48829 * std::cos(x) + 0.14128 * std::cos(2 * x) - 0.01168 * std::cos(3 * x));
    }

  for (int n = 0; n < N; n++) h->window_power += (double)win[n] * (double)win[n];

  // segments grouped by band, table order kept inside a band (the reference sums CH1's two runs
  // in table order, CE_Predictive_Node.cpp:173-179)
  std::vector<int> seg_begin(cfg.n_bands + 1, 0), seg_lo, seg_hi, bins_begin(cfg.n_bands + 1, 0), bins;
  for (int b = 0; b < cfg.n_bands; b++) {
    seg_begin[b] = (int)seg_lo.size();
    bins_begin[b] = (int)bins.size();
    for (int s = 0; s < cfg.n_segs; s++)
      if (cfg.segs[s].band == b) {
        seg_lo.push_back(cfg.segs[s].lo);
        seg_hi.push_back(cfg.segs[s].hi);
        for (int k = cfg.segs[s].lo; k < cfg.segs[s].hi; k++) bins.push_back(k);
      }
  }
  {
    // bin k sits in register j R3 + d of its thread: d = k / 256, j = ((k % 256) / 16) mod J (crn_frame.h, pass 3)
    const int J = 16 / R3;
    h->acc_mask = 0;
    for (int sgi = 0; sgi < cfg.n_segs; sgi++)
      for (int k = cfg.segs[sgi].lo; k < cfg.segs[sgi].hi; k++) h->acc_mask |= 1u << ((((k & 255) >> 4) % J) * R3 + (k >> 8));
  }
  if (N == 4096 && cfg.n_segs == cfg.n_bands && N % cfg.n_bands == 0 && cfg.decide != CRN_DECIDE_ANN) {
    const int W = N / cfg.n_bands;
    bool ok = W == 64 || W == 128 || W == 256;
    for (int b = 0; ok && b < cfg.n_bands; b++)
      ok = cfg.segs[b].band == b && cfg.segs[b].lo == b * W && cfg.segs[b].hi == (b + 1) * W;
    if (ok) h->aligned_shift = W == 64 ? 6 : W == 128 ? 7 : 8;
  }
  seg_begin[cfg.n_bands] = (int)seg_lo.size();
  bins_begin[cfg.n_bands] = (int)bins.size();
  if (bins.empty()) bins.push_back(0);

  // packed band table for the kernel's LDS copy (layout: crn_kernels.h)
  std::vector<int> band_tab(crn::kBandTabWords, 0);
  for (size_t i = 0; i < seg_begin.size(); i++) band_tab[i] = seg_begin[i];
  for (size_t i = 0; i < seg_lo.size(); i++) {
    band_tab[96 + i] = seg_lo[i];
    band_tab[256 + i] = seg_hi[i];
  }
  std::memcpy(&band_tab[416], cfg.thresh, sizeof(float) * CRN_MAX_BANDS);
  std::memcpy(&band_tab[544], cfg.ann_w_ih, sizeof(cfg.ann_w_ih));  // 30 doubles
  std::memcpy(&band_tab[604], cfg.ann_w_ho, sizeof(cfg.ann_w_ho));  // 24 doubles
  // Row entries for the register-resident band sums (epoch_close): every thread's accumulators sit
  // at bins base + 256 d, so a segment is cut at the 256-bin rows and each piece becomes
  // (row d, band, [lo, hi) inside the row), grouped by row, band-table order kept inside a row.
  // Only small plans qualify (<= 16 bands, <= 32 / R3 pieces per row); the others keep the LDS walk.
  h->n_row_entries = 0;
  {
    struct Piece { int d, band, lo, hi; };
    std::vector<Piece> pieces;
    for (int b = 0; b < cfg.n_bands; b++)
      for (int sg = seg_begin[b]; sg < seg_begin[b + 1]; sg++)
        for (int d = seg_lo[sg] >> 8; seg_lo[sg] < seg_hi[sg] && d <= (seg_hi[sg] - 1) >> 8; d++) {
          const int lo = std::max(seg_lo[sg], 256 * d) - 256 * d, hi = std::min(seg_hi[sg], 256 * (d + 1)) - 256 * d;
          pieces.push_back({d, b, lo, hi});
        }
    // fixed layout, no walk: row d owns words [512 + d * cap, 512 + (d + 1) * cap), cap = 32 / R3;
    // an unused slot is 0 (span 0)
    const int cap = crn::kRowEntryWords / R3;
    bool fits = cfg.n_bands <= 16 && !pieces.empty();
    std::vector<int> used(16, 0);
    for (const Piece &pc : pieces)
      if (++used[pc.d] > cap) fits = false;
    if (fits) {
      std::fill(used.begin(), used.end(), 0);
      for (const Piece &pc : pieces) band_tab[512 + pc.d * cap + used[pc.d]++] = (pc.band << 18) | (pc.lo << 9) | pc.hi;
      h->n_row_entries = (int)pieces.size();
    }
  }

  // twice the signed centre of every band (bins >= N / 2 are negative frequencies; lowest + highest signed bin, so a band with a
  // small gap in it — the reference plan's CH1 skips bins -1, -2 — is centred on its span): the carrier of the generator's
  // modulated signal kinds
  std::vector<int> band_c2(std::max(cfg.n_bands, 1), 0);
  for (int b = 0; b < cfg.n_bands; b++) {
    int lo = N, hi = -N;
    for (int i = bins_begin[b]; i < bins_begin[b + 1]; i++) {
      const int k = bins[i] >= N / 2 ? bins[i] - N : bins[i];
      lo = std::min(lo, k);
      hi = std::max(hi, k);
    }
    band_c2[b] = bins_begin[b + 1] > bins_begin[b] ? lo + hi : 0;
  }

  struct Piece { const void *src; size_t bytes; size_t off; };
  std::vector<Piece> pieces = {
      {tw1.data(), tw1.size() * sizeof(float2), 0},
      {tw2.data(), tw2.size() * sizeof(float2), 0},
      {win.data(), win.size() * sizeof(float), 0},
      {cfg.thresh, sizeof(float) * CRN_MAX_BANDS, 0},
      {seg_begin.data(), seg_begin.size() * sizeof(int), 0},
      {seg_lo.data(), seg_lo.size() * sizeof(int), 0},
      {seg_hi.data(), seg_hi.size() * sizeof(int), 0},
      {bins_begin.data(), bins_begin.size() * sizeof(int), 0},
      {bins.data(), bins.size() * sizeof(int), 0},
      {cfg.ann_w_ih, sizeof(cfg.ann_w_ih), 0},
      {cfg.ann_w_ho, sizeof(cfg.ann_w_ho), 0},
      {band_tab.data(), band_tab.size() * sizeof(int), 0},
      {band_c2.data(), band_c2.size() * sizeof(int), 0},
  };
  size_t total = 0;
  for (auto &p : pieces) {
    p.off = total;
    total = align_up(total + p.bytes, 256);
  }
  std::vector<char> host(total, 0);
  for (auto &p : pieces) std::memcpy(host.data() + p.off, p.src, p.bytes);
  void *slab = nullptr;
  hipError_t e = hipMalloc(&slab, total);
  if (e != hipSuccess) {
    undo();
    return crn::fail(CRN_ERR_NOMEM, std::string("hipMalloc(tables): ") + hipGetErrorString(e));
  }
  e = hipMemcpy(slab, host.data(), total, hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    (void)hipFree(slab);
    undo();
    return crn::fail(CRN_ERR_DEVICE, std::string("hipMemcpy(tables): ") + hipGetErrorString(e));
  }
  // hipFree waits for the device: launches still reading the previous slab (crn_sense_set_bands on a live handle) finish first
  if (h->d_tables) (void)hipFree(h->d_tables);
  h->d_tables = slab;
  char *base = static_cast<char *>(h->d_tables);
  h->d_tw1 = reinterpret_cast<const float2 *>(base + pieces[0].off);
  h->d_tw2 = reinterpret_cast<const float2 *>(base + pieces[1].off);
  h->d_window = reinterpret_cast<const float *>(base + pieces[2].off);
  h->d_thresh = reinterpret_cast<const float *>(base + pieces[3].off);
  h->d_band_seg_begin = reinterpret_cast<const int *>(base + pieces[4].off);
  h->d_seg_lo = reinterpret_cast<const int *>(base + pieces[5].off);
  h->d_seg_hi = reinterpret_cast<const int *>(base + pieces[6].off);
  h->d_band_bins_begin = reinterpret_cast<const int *>(base + pieces[7].off);
  h->d_band_bins = reinterpret_cast<const int *>(base + pieces[8].off);
  h->d_wih = reinterpret_cast<const double *>(base + pieces[9].off);
  h->d_who = reinterpret_cast<const double *>(base + pieces[10].off);
  h->d_band_tab = reinterpret_cast<const int *>(base + pieces[11].off);
  h->d_band_c2 = reinterpret_cast<const int *>(base + pieces[12].off);
  return CRN_OK;
}

extern "C" {

const char *crn_last_error(void) { return crn::g_err.c_str(); }
int crn_abi_version(void) { return CRN_ABI_VERSION; }

int crn_build_info(int32_t *built_hip, int32_t *runtime_hip) {
  const int built = HIP_VERSION;   // hip/hip_version.h of the toolchain that compiled this file
  int rt = 0;
  if (hipRuntimeGetVersion(&rt) != hipSuccess) rt = 0;
  if (built_hip) *built_hip = built;
  if (runtime_hip) *runtime_hip = rt;
  if (rt == 0) return crn::fail(CRN_ERR_STATE, "crn_build_info: no HIP runtime answers");
  if (rt / 10000000 != built / 10000000 || rt < 70000000)
    return crn::fail(CRN_ERR_STATE, "libcrnsense was built with HIP " + std::to_string(built) + " and needs a HIP runtime of the same major "
                                    "version, ROCm 7.0 or newer (gfx950); this machine's reports " + std::to_string(rt));
  return CRN_OK;
}

int crn_sense_create(const crn_cfg *cfg, crn_handle **out) {
  if (!out) return crn::fail(CRN_ERR_ARG, "crn_sense_create: null out");
  *out = nullptr;
  if (int rc = validate(cfg)) return rc;
  int ndev = 0;
  HIP_TRY(hipGetDeviceCount(&ndev));
  if (ndev < 1) return crn::fail(CRN_ERR_DEVICE, "no HIP device visible (libcrnsense has no CPU path)");
  if (cfg->device < 0 || cfg->device >= ndev) return crn::fail(CRN_ERR_ARG, "cfg.device out of range");
  HIP_TRY(hipSetDevice(cfg->device));

  crn_handle *h = new (std::nothrow) crn_handle(cfg->device);
  if (!h) return crn::fail(CRN_ERR_NOMEM, "out of host memory");
  h->cfg = *cfg;
  {
    int cus = 0, lds = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, cfg->device) == hipSuccess && cus > 0) h->n_cus = cus;
    if (hipDeviceGetAttribute(&lds, hipDeviceAttributeMaxSharedMemoryPerBlock, cfg->device) == hipSuccess && lds > 0) h->lds_budget = (size_t)lds;
  }
  if (int rc = build_tables(h)) {
    delete h;
    return rc;
  }
  hipError_t e = hipHostMalloc(reinterpret_cast<void **>(&h->upd), sizeof(crn_handle::UpdateSlot) * crn_handle::kUpdateSlots, hipHostMallocDefault);
  for (int i = 0; i < crn_handle::kUpdateSlots && e == hipSuccess; i++) e = hipEventCreateWithFlags(&h->upd_done[i], hipEventDisableTiming);
  if (e != hipSuccess) {
    (void)crn_sense_destroy(h);
    return crn::fail(CRN_ERR_NOMEM, std::string("crn_sense_create(update staging): ") + hipGetErrorString(e));
  }
  *out = h;
  return CRN_OK;
}

int crn_sense_destroy(crn_handle *h) {
  if (!h) return CRN_OK;
  // a ring's launcher thread launches through this handle and crn_ingest_destroy detaches from it: rings go first
  if (h->n_rings.load(std::memory_order_acquire) > 0)
    return crn::fail(CRN_ERR_STATE, "crn_sense_destroy: " + std::to_string(h->n_rings.load()) + " ingest ring(s) are still attached to this "
                                    "handle (crn_ingest_destroy them first)");
  (void)hipSetDevice(h->device);
  if (h->upd) (void)hipHostFree(h->upd);
  for (int i = 0; i < crn_handle::kUpdateSlots; i++)
    if (h->upd_done[i]) (void)hipEventDestroy(h->upd_done[i]);
  if (h->h_nf_features) (void)hipHostFree(h->h_nf_features);
  if (h->d_nf_features) (void)hipFree(h->d_nf_features);
  if (h->d_scratch) (void)hipFree(h->d_scratch);
  if (h->h_results) (void)hipHostFree(h->h_results);
  if (h->h_small) (void)hipHostFree(h->h_small);
  if (h->d_tables) (void)hipFree(h->d_tables);
  if (h->d_nf_scratch) (void)hipFree(h->d_nf_scratch);
  for (int i = 0; i < crn_handle::kTimedSlots; i++) {
    if (h->t_start[i]) (void)hipEventDestroy(h->t_start[i]);
    if (h->t_stop[i]) (void)hipEventDestroy(h->t_stop[i]);
  }
  delete h;
  return CRN_OK;
}

// internal (crn_ingest.cpp): a ring attaches to / detaches from its handle
int crn_sense_ring_count(crn_handle *h, int delta) {
  if (!h) return crn::fail(CRN_ERR_ARG, "null handle");
  h->n_rings.fetch_add(delta, std::memory_order_acq_rel);
  return CRN_OK;
}

// internal (crn_ingest.cpp): an empty launch on the ring's stream, queued at the pre-wake of a batch that follows an idle stretch —
// the HIP calls of the first launch on a queue that sat idle for 100 ms take 20 us instead of 5 (tools/engine_idle_gap.py)
int crn_sense_warm_stream(crn_handle *h, void *stream) {
  if (!h) return crn::fail(CRN_ERR_ARG, "null handle");
  HIP_TRY(hipSetDevice(h->device));   // (the launcher thread, at every pre-wake: no lock, nothing of cfg)
  HIP_TRY(crn::launch_nop(static_cast<hipStream_t>(stream)));
  return CRN_OK;
}

// internal (crn_ingest.cpp): the configuration a handle was created with
int crn_sense_cfg_of(crn_handle *h, crn_cfg *out) {
  if (!h || !out) return crn::fail(CRN_ERR_ARG, "null handle");
  std::lock_guard<std::mutex> lk(h->tables_mu);
  *out = h->cfg;
  return CRN_OK;
}

int crn_sense_dealt_launches(crn_handle *h, int64_t *n) {
  if (!h || !n) return crn::fail(CRN_ERR_ARG, "null handle / counter");
  *n = h->n_dealt.load(std::memory_order_relaxed);
  return CRN_OK;
}

int crn_sense_set_variant(crn_handle *h, int32_t variant) {
  if (!h) return crn::fail(CRN_ERR_ARG, "null handle");
  std::lock_guard<std::mutex> lk(h->tables_mu);
  if (variant >= 100 && variant <= 164) {  // A/B: 100 + n = n epoch groups per workgroup (100 = automatic)
    h->groups_per_wg = variant - 100;
    return CRN_OK;
  }
  if (variant >= 200 && variant <= 264) {  // A/B: 200 + n = n x 256 epoch groups in the short tail workgroups
    h->tail_groups = (int64_t)(variant - 200) * 256;
    return CRN_OK;
  }
  if (variant >= 300 && variant <= 364) {  // A/B: 300 + n = n epoch groups per tail workgroup (300 = automatic)
    h->tail_groups_per_wg = variant - 300;
    return CRN_OK;
  }
  if (variant >= 400 && variant <= 402) {  // A/B: the dealt-frame kernel of small launches: 400 automatic, 401 never, 402 at any batch size
    h->deal_max_epochs = variant == 400 ? -1 : variant == 401 ? 0 : (int64_t)0x7fffffff;
    return CRN_OK;
  }
  if (variant < 0 || variant > crn::sense_num_variants()) return crn::fail(CRN_ERR_ARG, "variant out of range");
  if (!crn::sense_variant_available(variant))
    return crn::fail(CRN_ERR_ARG, "variant " + std::to_string(variant) + " is not a form of this library: the measurement variants (7, 17, 19-22, 26, 27) "
                                  "are compiled into libcrnsense_ab.so (make -C csrc ab), not into the shipped library; any other number names "
                                  "a form that no longer exists");
  h->variant = variant;
  return CRN_OK;
}

int crn_sense_kernel_info(crn_handle *h, char *name, int32_t name_len, int32_t *threads_per_block,
                          int32_t *lds_bytes, int32_t *epochs_per_block) {
  if (!h) return crn::fail(CRN_ERR_ARG, "null handle");
  std::lock_guard<std::mutex> lk(h->tables_mu);
  int thr = 0, lds = 0, epb = 0;
  crn::sense_geometry(h->cfg.fft_len, h->variant, &thr, &lds, &epb);
  if (threads_per_block) *threads_per_block = thr;
  if (lds_bytes) *lds_bytes = lds;
  if (epochs_per_block) *epochs_per_block = epb;
  if (name && name_len > 0) {
    int nbuf = 1, pf = 0, nt = 0, tl = 0, pk = 0;
    crn::sense_variant(h->cfg.fft_len, h->cfg.mode == CRN_MODE_REF_MAG || h->cfg.window != CRN_WINDOW_RECT ? -1 : h->variant,
                       &nbuf, &pf, &nt, &tl, &pk);
    const bool plain4096 = h->cfg.fft_len == 4096 && h->cfg.mode != CRN_MODE_REF_MAG && h->cfg.window == CRN_WINDOW_RECT;
    // what a launch without a spectrum output runs (a spectrum request falls back to full rows / the LDS close)
    bool reg_close = h->n_row_entries > 0 && h->cfg.window == CRN_WINDOW_RECT;
    if (plain4096) {  // the forms of the plain kernel that carry the register close
      const int v = h->variant == 0 ? 13 : h->variant;
      const bool rows_ok = (h->acc_mask & ~0x8267u) == 0;
      reg_close = reg_close && (v == 2 || v == 13 || v == 17 || (v == 7 && rows_ok));
    }
    // pass 3 / accumulate pruned to the reference channel plan's registers: the plain 4096-point kernel's default form, and the
    // register-close kernels of every other size and mode (what a launch without a spectrum output runs)
    const unsigned ref_mask = crn::sense_ref_acc_mask(h->cfg.fft_len);
    const bool inside = (h->acc_mask & ~ref_mask) == 0 && ref_mask != 0xFFFFu && h->variant != 2;
    const bool pruned = reg_close && inside && h->cfg.window == CRN_WINDOW_RECT && (!plain4096 || h->variant == 0 || h->variant == 13);
    // periodic Hann in energy mode: the window is folded into pass 1 (whole frames); with the Welch scan's plan
    // (N = 4096, equal contiguous bands) the close forms band sums by DPP
    const bool hann_fold = h->cfg.window == CRN_WINDOW_HANN && h->cfg.mode != CRN_MODE_REF_MAG;
    const bool aligned = hann_fold && h->cfg.fft_len == 4096 && h->aligned_shift != 0;
    if (hann_fold) tl = 1;
    char prune_note[64] = "";
    if (pruned) std::snprintf(prune_note, sizeof(prune_note), ",PASS3_ROWS=%d-of-16(reference channel plan)", __builtin_popcount(ref_mask));
    std::snprintf(name, (size_t)name_len, "sense_kernel<R3=%d,NBUF=%d,PREFETCH=%d,NT=%d,TW2LDS=%d,PK=%d,MAG=%d,WIN=%s,CLOSE=%s%s>",
                  h->cfg.fft_len / 256, nbuf, pf, nt, tl, pk,
                  h->cfg.mode == CRN_MODE_REF_MAG, h->cfg.window == CRN_WINDOW_RECT ? "0" : hann_fold ? "hann-in-pass1" : "table",
                  aligned ? "aligned-bands(dpp)" : reg_close ? "registers" : "lds",
                  prune_note);
  }
  return CRN_OK;
}

static int resolve_strides(const crn_handle *h, int32_t L, int64_t *epoch_stride, int *frame_stride) {
  const crn_cfg &c = h->cfg;
  if (L < 1 || L > c.fft_len)
    return crn::fail(CRN_ERR_ARG, "samples_per_frame must be in 1..fft_len (the reference's unchecked memcpy, "
                                   "CE_Predictive_Node.cpp:149, is rejected here)");
  if (c.hop != c.fft_len && L != c.fft_len)
    return crn::fail(CRN_ERR_ARG, "overlapped frames (hop < fft_len) need samples_per_frame == fft_len");
  *frame_stride = c.hop == c.fft_len ? L : c.hop;
  if (*epoch_stride <= 0) *epoch_stride = (int64_t)c.frames_per_epoch * *frame_stride;
  return CRN_OK;
}

namespace {
// Collect the durations of timed launches that have finished (timing_mu held); `wait`: also the one occupying slot `must_free`.
void collect_timings(crn_handle *h, bool wait_oldest) {
  while (h->t_collected < h->t_issued) {
    const int s = (int)(h->t_collected % crn_handle::kTimedSlots);
    hipError_t q = hipEventQuery(h->t_stop[s]);
    if (q == hipErrorNotReady && wait_oldest) q = hipEventSynchronize(h->t_stop[s]);
    wait_oldest = false;
    if (q == hipErrorNotReady) break;
    float ms = 0.f;
    if (q == hipSuccess && hipEventElapsedTime(&ms, h->t_start[s], h->t_stop[s]) == hipSuccess) {
      const int64_t n = h->t_collected - h->t_dropped;
      h->kernel_ms += ms;
      h->kernel_ms_last = ms;
      h->kernel_ms_min = n == 0 ? ms : std::min(h->kernel_ms_min, (double)ms);
      h->kernel_ms_max = n == 0 ? ms : std::max(h->kernel_ms_max, (double)ms);
    } else {
      h->t_dropped++;   // a launch that failed on the device: no duration
    }
    h->t_collected++;
  }
}
}  // namespace

int crn_sense_set_timing(crn_handle *h, int32_t on) {
  if (!h) return crn::fail(CRN_ERR_ARG, "null handle");
  std::lock_guard<std::mutex> lk(h->timing_mu);
  if (on && !h->t_start[0]) {
    HIP_TRY(hipSetDevice(h->device));
    for (int i = 0; i < crn_handle::kTimedSlots; i++) {
      HIP_TRY(hipEventCreate(&h->t_start[i]));
      HIP_TRY(hipEventCreate(&h->t_stop[i]));
    }
  }
  h->timing = on != 0;
  return CRN_OK;
}

int crn_sense_get_stats(crn_handle *h, crn_sense_stats *out) {
  if (!h || !out) return crn::fail(CRN_ERR_ARG, "null handle / stats");
  std::lock_guard<std::mutex> lk(h->timing_mu);
  collect_timings(h, false);
  out->launches = h->n_launches.load(std::memory_order_relaxed);
  out->epochs = h->n_epochs.load(std::memory_order_relaxed);
  out->samples = h->n_samples.load(std::memory_order_relaxed);
  out->timed_launches = h->t_collected - h->t_dropped;
  out->kernel_ms = h->kernel_ms;
  out->kernel_ms_last = h->kernel_ms_last;
  out->kernel_ms_min = h->kernel_ms_min;
  out->kernel_ms_max = h->kernel_ms_max;
  return CRN_OK;
}

static int run_device_impl(crn_handle *h, const void *d_iq, int64_t n_epochs, int32_t samples_per_frame,
                           int64_t epoch_stride, const crn_out *d_out, void *stream, bool sc16) {
  if (!h || !d_out) return crn::fail(CRN_ERR_ARG, "null handle / outputs");
  if (n_epochs < 0) return crn::fail(CRN_ERR_ARG, "n_epochs < 0");
  if (n_epochs == 0) return CRN_OK;
  if (!d_iq) return crn::fail(CRN_ERR_ARG, "null IQ pointer");
  const int64_t sample_bytes = sc16 ? 4 : 8;
  if ((reinterpret_cast<uintptr_t>(d_iq) & (uintptr_t)(sample_bytes - 1)) != 0)
    return crn::fail(CRN_ERR_ARG, sc16 ? "IQ pointer must be 4-byte aligned" : "IQ pointer must be 8-byte aligned");
  // one plan, whole, from here until the kernel is enqueued (live updates from another thread wait; see crn_handle::tables_mu)
  std::lock_guard<std::mutex> tables_lk(h->tables_mu);
  int frame_stride = 0;
  if (int rc = resolve_strides(h, samples_per_frame, &epoch_stride, &frame_stride)) return rc;
  if (n_epochs > (int64_t)0x7fffffff) return crn::fail(CRN_ERR_ARG, "n_epochs too large for one launch");
  HIP_TRY(hipSetDevice(h->device));  // a NULL stream / a launch follows the calling thread's current device
  // a workgroup addresses its window with 32-bit byte offsets
  if ((64 * epoch_stride + (int64_t)(h->cfg.frames_per_epoch + 1) * frame_stride + 2 * (int64_t)h->cfg.fft_len) * sample_bytes >= ((int64_t)1 << 31))
    return crn::fail(CRN_ERR_ARG, "epoch_stride too large (a workgroup window must stay below 2 GiB)");
  const crn_cfg &c = h->cfg;
  crn::SenseParams p{};
  p.iq = reinterpret_cast<const float2 *>(d_iq);   // int16 pairs when sc16: the kernel instantiation knows
  p.n_epochs = n_epochs;
  p.epoch_stride = epoch_stride;
  p.total_samples = (n_epochs - 1) * epoch_stride + (int64_t)(c.frames_per_epoch - 1) * frame_stride +
                    (c.hop == c.fft_len ? samples_per_frame : c.fft_len);
  p.frame_stride = frame_stride;
  p.L = samples_per_frame;
  p.K = c.frames_per_epoch;
  {
    // Several epoch groups per workgroup amortise its prologue (twiddles and tables loaded once, the next
    // epoch's first frame in flight across the close); the single-group workgroups at the end keep the drain
    // short, so two rounds of big workgroups over the 256 CUs x 4 slots are enough (measured at N = 4096:
    // +0.5-1.5 % on 2048 .. 12288-epoch batches over the earlier n_groups / 4096).
    const int groups = 256 / (c.fft_len / 16);
    const int64_t n_groups = (n_epochs + groups - 1) / groups;
    // workgroup slots of this device: CUs x workgroups per CU (4 for the plain kernels' register budget, 3 for the windowed ones)
    const int64_t slots4 = (int64_t)h->n_cus * 4, slots3 = (int64_t)h->n_cus * 3;
    int64_t epw = n_groups / (2 * slots4);
    epw = epw < 1 ? 1 : epw > 4 ? 4 : epw;
    // the last `tail` groups go to short workgroups (dispatched last): a short drain
    int64_t tail = slots4;      // one single-group workgroup per workgroup slot: +0.9 % at N = 4096
    int64_t tail_epw = 1;
    // The Welch stream (windowed, hop = N/2, dense epochs) reads one half-frame twice per workgroup span — a span's first
    // half-frame is the previous span's last — so its spans are made long: ~256 frames per big workgroup while at least ~2.7
    // rounds of them remain over the 768 slots (3 workgroups per CU), a quarter of that per tail workgroup, one tail
    // workgroup per slot.  At K = 8 that is 32 epochs / 8 epochs / 6144 epochs: traffic 1.005 x the algorithmic bytes instead
    // of 1.033 x with 4-epoch spans and a single-epoch tail, and 1-2 % less time (profiles/r05_welch_spans.txt).
    if (c.window != CRN_WINDOW_RECT && c.hop * 2 == c.fft_len && epoch_stride == (int64_t)c.frames_per_epoch * c.hop) {
      const int64_t slots = slots3;
      epw = std::min<int64_t>(std::max<int64_t>(256 / c.frames_per_epoch, 1), n_groups * 3 / (8 * slots));   // >= 2.67 rounds of them
      epw = epw < 1 ? 1 : epw > 64 ? 64 : epw;
      tail_epw = epw / 4 < 1 ? 1 : epw / 4 > 8 ? 8 : epw / 4;
      tail = slots * tail_epw;
    }
    p.groups_per_wg = (int)(h->groups_per_wg > 0 ? h->groups_per_wg : epw);
    if (h->tail_groups >= 0) tail = h->tail_groups;
    if (tail > n_groups / 4) tail = n_groups / 4;
    p.tail_groups_per_wg = (int)(h->tail_groups_per_wg > 0 ? h->tail_groups_per_wg : tail_epw);
    p.n_big_wgs = (n_groups - tail) / p.groups_per_wg;
  }
  {
    // A launch of a few epochs (the engine's: one) leaves most of every workgroup idle in the streaming kernel — an epoch is one lane
    // group running its K frames one after the other.  Up to one epoch per compute unit the dealt-frame kernel spreads an epoch's
    // frames over the lane groups of a workgroup of its own instead (csrc/crn_sense_kernel.h: sense_kernel_dealt; same results bit
    // for bit).  Measured (profiles/r05_dealt_frames_ab.txt, us per launch, streaming -> dealt): 1 reference epoch 19.7 -> 10.0 from
    // HBM and 29.4 -> 15.4 from pinned host memory (the ring's launch); 256 epochs 20.3 -> 10.9; from 512 epochs on — two
    // workgroups per CU — the energy forms lose (17.5 -> 20.5), so the switch sits at one per CU.
    const int64_t deal_max = h->deal_max_epochs >= 0 ? h->deal_max_epochs : (int64_t)h->n_cus;
    if (n_epochs <= deal_max && (h->variant == 0 || h->variant == 13))
      p.deal_rounds = crn::sense_deal_rounds(c.fft_len, c.mode == CRN_MODE_REF_MAG, c.window != CRN_WINDOW_RECT,
                                             c.window == CRN_WINDOW_HANN && samples_per_frame == c.fft_len, c.frames_per_epoch, h->lds_budget);
  }
  p.tw1 = h->d_tw1;
  p.tw2 = h->d_tw2;
  p.window = h->d_window;
  p.band_tab = h->d_band_tab;
  p.band_seg_begin = h->d_band_seg_begin;
  p.seg_lo = h->d_seg_lo;
  p.seg_hi = h->d_seg_hi;
  p.thresh = h->d_thresh;
  p.ann_w_ih = h->d_wih;
  p.ann_w_ho = h->d_who;
  p.ann_threshold = c.ann_threshold;
  p.n_bands = c.n_bands;
  p.decide = c.decide;
  p.ref_band = c.ref_band;
  p.hann_sym = c.window == CRN_WINDOW_HANN;
  p.aligned_shift = d_out->spectrum == nullptr ? h->aligned_shift : 0;
  p.acc_mask = h->variant == 2 ? 0xFFFFu : h->acc_mask;   // variant 2: no pruning at any size
  {  // 1 / full scale for a sum of magnitudes, its square for energies (2^-15 / 2^-30 by default: exact)
    const double u = 1.0 / h->wire_full_scale;
    p.wire_unscale = (float)(c.mode == CRN_MODE_REF_MAG ? u : u * u);
  }
  p.n_row_entries = h->n_row_entries;
  p.features = d_out->features;
  // (a measurement form that writes time stamps puts them there: libcrnsense_ab.so only)
  p.ann_out = (c.decide == CRN_DECIDE_ANN || crn::sense_variant_traces(h->variant)) ? d_out->ann_out : nullptr;
  p.decision = d_out->decision;
  p.occupancy = d_out->occupancy;
  p.spectrum = d_out->spectrum;
  int slot = -1;
  {
    std::lock_guard<std::mutex> lk(h->timing_mu);
    if (h->timing) {
      if (h->t_issued - h->t_collected == crn_handle::kTimedSlots) collect_timings(h, true);   // 16 launches behind: wait for the oldest
      slot = (int)(h->t_issued++ % crn_handle::kTimedSlots);
      HIP_TRY(hipEventRecord(h->t_start[slot], static_cast<hipStream_t>(stream)));
    }
  }
  int deal_rounds_run = 0;   // the form really launched: 0 when the device refused the dealt form's LDS and the streaming kernel took it
  const hipError_t le = crn::launch_sense(p, c.fft_len, c.mode == CRN_MODE_REF_MAG, c.window != CRN_WINDOW_RECT, h->variant,
                                          static_cast<hipStream_t>(stream), sc16, &deal_rounds_run);
  if (slot >= 0) (void)hipEventRecord(h->t_stop[slot], static_cast<hipStream_t>(stream));   // also after a failed launch: the slot must complete
  if (le == hipErrorNotSupported && sc16)
    return crn::fail(CRN_ERR_ARG, "this library was built without the wire-format kernels (make -C csrc SC16=1 builds libcrnsense_sc16.so)");
  if (le != hipSuccess) return crn::fail(CRN_ERR_DEVICE, std::string("launch_sense: ") + hipGetErrorString(le));
  // every input sample once: consecutive epochs closer together than an epoch is long (Welch) share their overlap
  const int64_t extent = (int64_t)(c.frames_per_epoch - 1) * frame_stride + (c.hop == c.fft_len ? samples_per_frame : c.fft_len);
  h->n_launches.fetch_add(1, std::memory_order_relaxed);
  if (deal_rounds_run > 0) h->n_dealt.fetch_add(1, std::memory_order_relaxed);
  h->n_epochs.fetch_add(n_epochs, std::memory_order_relaxed);
  h->n_samples.fetch_add((n_epochs - 1) * std::min(epoch_stride, extent) + extent, std::memory_order_relaxed);
  return CRN_OK;
}

int crn_sense_run_device(crn_handle *h, const float *d_iq, int64_t n_epochs, int32_t samples_per_frame,
                         int64_t epoch_stride, const crn_out *d_out, void *stream) {
  return run_device_impl(h, d_iq, n_epochs, samples_per_frame, epoch_stride, d_out, stream, false);
}

// internal (crn_ingest.cpp): either sample format through one call — a ring of wire-format packets (bytes_per_sample 4) exists only
// in a library built with the wire-format kernels
int crn_sense_run_device_any(crn_handle *h, const void *d_iq, int32_t bytes_per_sample, int64_t n_epochs, int32_t samples_per_frame,
                             int64_t epoch_stride, const crn_out *d_out, void *stream) {
  return run_device_impl(h, d_iq, n_epochs, samples_per_frame, epoch_stride, d_out, stream, bytes_per_sample == 4);
}

#ifdef CRN_WITH_SC16   // optional: wire-format input (make SC16=1 -> libcrnsense_sc16.so; include/crn_sense_sc16.h)
int crn_sense_run_device_sc16(crn_handle *h, const int16_t *d_iq, int64_t n_epochs, int32_t samples_per_frame,
                              int64_t epoch_stride, const crn_out *d_out, void *stream) {
  return run_device_impl(h, d_iq, n_epochs, samples_per_frame, epoch_stride, d_out, stream, true);
}

int crn_sense_set_wire_full_scale(crn_handle *h, double full_scale) {
  if (!h) return crn::fail(CRN_ERR_ARG, "null handle");
  if (!(full_scale >= 1.0 && full_scale <= 65536.0)) return crn::fail(CRN_ERR_ARG, "full_scale must be in 1..65536");
  std::lock_guard<std::mutex> lk(h->tables_mu);
  h->wire_full_scale = full_scale;
  return CRN_OK;
}

int crn_pack_sc16_device(crn_handle *h, const float *d_iq, int64_t n_samples, int16_t *d_out, void *stream) {
  if (!h || !d_iq || !d_out) return crn::fail(CRN_ERR_ARG, "null handle / buffer");
  if (n_samples < 0) return crn::fail(CRN_ERR_ARG, "n_samples < 0");
  std::lock_guard<std::mutex> lk(h->tables_mu);
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(crn::launch_pack_sc16(d_iq, n_samples, d_out, (float)h->wire_full_scale, static_cast<hipStream_t>(stream)));
  return CRN_OK;
}
#endif  // CRN_WITH_SC16

int crn_sense_run_host(crn_handle *h, const float *iq, int64_t n_epochs, int32_t samples_per_frame,
                       int64_t epoch_stride, const crn_out *out) {
  if (!h || !out) return crn::fail(CRN_ERR_ARG, "null handle / outputs");
  if (n_epochs < 0) return crn::fail(CRN_ERR_ARG, "n_epochs < 0");
  if (n_epochs == 0) return CRN_OK;
  if (!iq) return crn::fail(CRN_ERR_ARG, "null IQ pointer");
  int frame_stride = 0;
  if (int rc = resolve_strides(h, samples_per_frame, &epoch_stride, &frame_stride)) return rc;
  const crn_cfg &c = h->cfg;
  HIP_TRY(hipSetDevice(h->device));
  // samples touched: last epoch start + (K-1) frame strides + the last frame
  const int64_t last_frame = c.hop == c.fft_len ? samples_per_frame : c.fft_len;
  const size_t n_samples = (size_t)((n_epochs - 1) * epoch_stride + (int64_t)(c.frames_per_epoch - 1) * frame_stride + last_frame);
  const size_t b_iq = align_up(n_samples * 8, 256);
  const size_t b_feat = align_up((size_t)n_epochs * c.n_bands * sizeof(float), 256);
  const size_t b_ann = align_up((size_t)n_epochs * 3 * sizeof(double), 256);
  const size_t b_dec = align_up((size_t)n_epochs * sizeof(int32_t), 256);
  const size_t b_occ = align_up((size_t)n_epochs * c.n_bands, 256);
  const size_t b_spec = out->spectrum ? align_up((size_t)n_epochs * c.fft_len * sizeof(float), 256) : 0;
  const size_t need = b_iq + b_feat + b_ann + b_dec + b_occ + b_spec;
  const size_t res_bytes = b_feat + b_ann + b_dec + b_occ;
  if (!out->spectrum && n_samples * 8 <= kInPlaceBytes) {
    // A decision's worth of samples (the engine's synchronous form: one epoch of 10 x 512): staged in pinned memory that the
    // kernel reads, and whose tail it writes the results to, over the bus itself — one launch and one wait instead of upload +
    // launch + download.
    if (b_iq + res_bytes > h->h_small_bytes) {
      if (h->h_small) (void)hipHostFree(h->h_small);
      h->h_small = nullptr;
      h->h_small_bytes = 0;
      hipError_t e = hipHostMalloc(&h->h_small, b_iq + res_bytes, hipHostMallocDefault);
      if (e != hipSuccess) return crn::fail(CRN_ERR_NOMEM, std::string("hipHostMalloc(in-place staging): ") + hipGetErrorString(e));
      h->h_small_bytes = b_iq + res_bytes;
    }
    char *b = static_cast<char *>(h->h_small);
    std::memcpy(b, iq, n_samples * 8);
    crn_out d{};
    d.features = reinterpret_cast<float *>(b + b_iq);
    d.ann_out = reinterpret_cast<double *>(b + b_iq + b_feat);
    d.decision = reinterpret_cast<int32_t *>(b + b_iq + b_feat + b_ann);
    d.occupancy = reinterpret_cast<uint8_t *>(b + b_iq + b_feat + b_ann + b_dec);
    if (int rc = crn_sense_run_device(h, reinterpret_cast<const float *>(b), n_epochs, samples_per_frame, epoch_stride, &d, nullptr)) return rc;
    HIP_TRY(hipStreamSynchronize(nullptr));
    const char *r = b + b_iq;
    if (out->features) std::memcpy(out->features, r, (size_t)n_epochs * c.n_bands * sizeof(float));
    if (out->ann_out && c.decide == CRN_DECIDE_ANN) std::memcpy(out->ann_out, r + b_feat, (size_t)n_epochs * 3 * sizeof(double));
    if (out->decision) std::memcpy(out->decision, r + b_feat + b_ann, (size_t)n_epochs * sizeof(int32_t));
    if (out->occupancy) std::memcpy(out->occupancy, r + b_feat + b_ann + b_dec, (size_t)n_epochs * c.n_bands);
    return CRN_OK;
  }
  if (need > h->scratch_bytes) {
    if (h->d_scratch) (void)hipFree(h->d_scratch);
    h->d_scratch = nullptr;
    h->scratch_bytes = 0;
    hipError_t e = hipMalloc(&h->d_scratch, need);
    if (e != hipSuccess) return crn::fail(CRN_ERR_NOMEM, std::string("hipMalloc(scratch): ") + hipGetErrorString(e));
    h->scratch_bytes = need;
  }
  char *b = static_cast<char *>(h->d_scratch);
  float *d_iq = reinterpret_cast<float *>(b);
  crn_out d{};
  d.features = reinterpret_cast<float *>(b + b_iq);
  d.ann_out = reinterpret_cast<double *>(b + b_iq + b_feat);
  d.decision = reinterpret_cast<int32_t *>(b + b_iq + b_feat + b_ann);
  d.occupancy = reinterpret_cast<uint8_t *>(b + b_iq + b_feat + b_ann + b_dec);
  d.spectrum = out->spectrum ? reinterpret_cast<float *>(b + b_iq + b_feat + b_ann + b_dec + b_occ) : nullptr;
  hipStream_t s = nullptr;
  HIP_TRY(hipMemcpyAsync(d_iq, iq, n_samples * 8, hipMemcpyHostToDevice, s));
  if (int rc = crn_sense_run_device(h, d_iq, n_epochs, samples_per_frame, epoch_stride, &d, s)) return rc;
  // features | ann_out | decision | occupancy sit back to back in the scratch slab: one D2H into
  // pinned staging, then scatter on the host (a decision costs one upload, one launch, one download)
  if (res_bytes > h->h_results_bytes) {
    if (h->h_results) (void)hipHostFree(h->h_results);
    h->h_results = nullptr;
    h->h_results_bytes = 0;
    hipError_t e = hipHostMalloc(&h->h_results, res_bytes, hipHostMallocDefault);
    if (e != hipSuccess) return crn::fail(CRN_ERR_NOMEM, std::string("hipHostMalloc(results): ") + hipGetErrorString(e));
    h->h_results_bytes = res_bytes;
  }
  HIP_TRY(hipMemcpyAsync(h->h_results, b + b_iq, res_bytes, hipMemcpyDeviceToHost, s));
  if (out->spectrum) HIP_TRY(hipMemcpyAsync(out->spectrum, d.spectrum, (size_t)n_epochs * c.fft_len * sizeof(float), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  const char *r = static_cast<const char *>(h->h_results);
  if (out->features) std::memcpy(out->features, r, (size_t)n_epochs * c.n_bands * sizeof(float));
  if (out->ann_out && c.decide == CRN_DECIDE_ANN) std::memcpy(out->ann_out, r + b_feat, (size_t)n_epochs * 3 * sizeof(double));
  if (out->decision) std::memcpy(out->decision, r + b_feat + b_ann, (size_t)n_epochs * sizeof(int32_t));
  if (out->occupancy) std::memcpy(out->occupancy, r + b_feat + b_ann + b_dec, (size_t)n_epochs * c.n_bands);
  return CRN_OK;
}

int crn_sense_reserve_host(crn_handle *h, int64_t max_epochs, int32_t want_spectrum) {
  if (!h) return crn::fail(CRN_ERR_ARG, "null handle");
  if (max_epochs < 1) return crn::fail(CRN_ERR_ARG, "max_epochs < 1");
  const crn_cfg &c = h->cfg;
  // a run of zeros at the largest size: allocates the scratch slab and the pinned result staging,
  // loads the code object and sets the kernel's LDS attribute
  const int64_t stride = (int64_t)c.frames_per_epoch * c.hop;
  const size_t n_samples = (size_t)(max_epochs * stride + (c.fft_len - c.hop));
  std::vector<float> zeros(n_samples * 2, 0.f);
  std::vector<float> feat((size_t)max_epochs * c.n_bands), spec(want_spectrum ? (size_t)max_epochs * c.fft_len : 0);
  crn_out o{};
  o.features = feat.data();
  o.spectrum = want_spectrum ? spec.data() : nullptr;
  return crn_sense_run_host(h, zeros.data(), max_epochs, c.fft_len, 0, &o);
}

namespace {
// Updates copy from pinned staging with hipMemcpyAsync and mark their slot with an event; a stream that is being captured into a
// hipGraph would record both into the graph, where the event never completes for the host and every replay would upload whatever the
// slot holds by then.  Refused: make the update outside the capture (launches capture fine: tests/test_graph.py).
int refuse_capture(hipStream_t st, const char *what) {
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (st != nullptr && hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone)
    return crn::fail(CRN_ERR_STATE, std::string(what) + ": the stream is being captured into a hipGraph; updates cannot be captured (their pinned "
                                                       "staging slot is reused) — make them outside the capture");
  return CRN_OK;
}

// A pinned staging slot whose last copy has completed (`lk` = tables_mu, held).  Normally the first one tried; when all eight are still in
// flight the lock is RELEASED while this thread waits for the oldest — a launch on another thread never waits for an update's copy.
// *waited says that happened: whatever the caller checked under the lock before (the number of bands, the decision rule) may have been
// changed by a crn_sense_set_bands on another thread in that window, and the caller checks it again before it writes anything.
// (The event waited for may be re-recorded by another updater meanwhile: the wait is only a hint, the loop queries every slot afresh.)
int take_update_slot(crn_handle *h, std::unique_lock<std::mutex> &lk, int *slot, bool *waited) {
  *waited = false;
  for (;;) {
    for (int k = 0; k < crn_handle::kUpdateSlots; k++) {
      const int i = (int)((h->upd_next + k) % crn_handle::kUpdateSlots);
      if (h->upd_used[i]) {
        const hipError_t q = hipEventQuery(h->upd_done[i]);
        if (q == hipErrorNotReady) continue;
        if (q != hipSuccess) return crn::fail(CRN_ERR_DEVICE, std::string("hipEventQuery(update slot): ") + hipGetErrorString(q));
      }
      h->upd_used[i] = true;
      h->upd_next = i + 1;
      *slot = i;
      return CRN_OK;
    }
    const hipEvent_t oldest = h->upd_done[h->upd_next % crn_handle::kUpdateSlots];
    lk.unlock();
    const hipError_t e = hipEventSynchronize(oldest);
    lk.lock();
    *waited = true;
    if (e != hipSuccess) return crn::fail(CRN_ERR_DEVICE, std::string("hipEventSynchronize(update slot): ") + hipGetErrorString(e));
  }
}

// The reduction + its read-back (nf_mu held, tables_mu NOT held: the wait stalls nobody's launch).
int noise_floor_run(crn_handle *h, const float *d_features, int64_t n_epochs, int n_bands, float *nf_out, hipStream_t st) {
  if (!h->d_nf_scratch) HIP_TRY(hipMalloc(&h->d_nf_scratch, (crn::kNoiseFloorMaxEpochs + 1) * sizeof(float)));
  const int n = (int)std::min<int64_t>(n_epochs, crn::kNoiseFloorMaxEpochs);
  HIP_TRY(crn::launch_noise_floor(d_features, n, n_bands, h->d_nf_scratch, st));
  HIP_TRY(hipMemcpyAsync(nf_out, h->d_nf_scratch + crn::kNoiseFloorMaxEpochs, sizeof(float), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  return CRN_OK;
}

int bands_of(crn_handle *h) {
  std::lock_guard<std::mutex> lk(h->tables_mu);
  return h->cfg.n_bands;
}

int set_thresholds_locked(crn_handle *h, std::unique_lock<std::mutex> &lk, const float *thresh, int32_t n_bands, hipStream_t st) {
  int slot = 0;
  bool waited = false;
  if (int rc = take_update_slot(h, lk, &slot, &waited)) return rc;
  // (the slot taken stays marked used with its last, completed, event: the next update takes it)
  if (waited && n_bands != h->cfg.n_bands)
    return crn::fail(CRN_ERR_STATE, "the band plan changed while this update waited for a staging slot: nothing was written (set the thresholds of the new plan)");
  std::memcpy(h->cfg.thresh, thresh, sizeof(float) * (size_t)n_bands);
  float *src = h->upd[slot].thresh;
  std::memcpy(src, h->cfg.thresh, sizeof(float) * CRN_MAX_BANDS);
  // the two device copies the kernels read: the table and the packed band table's threshold words (layout: crn_kernels.h)
  HIP_TRY(hipMemcpyAsync(const_cast<float *>(h->d_thresh), src, sizeof(float) * CRN_MAX_BANDS, hipMemcpyHostToDevice, st));
  HIP_TRY(hipMemcpyAsync(const_cast<int *>(h->d_band_tab) + 416, src, sizeof(float) * CRN_MAX_BANDS, hipMemcpyHostToDevice, st));
  HIP_TRY(hipEventRecord(h->upd_done[slot], st));
  return CRN_OK;
}
}  // namespace

int crn_noise_floor_device(crn_handle *h, const float *d_features, int64_t n_epochs, float *nf_out, void *stream) {
  if (!h || !d_features || !nf_out) return crn::fail(CRN_ERR_ARG, "null handle / features / result");
  if (n_epochs < 1) return crn::fail(CRN_ERR_ARG, "n_epochs < 1");
  std::lock_guard<std::mutex> nf(h->nf_mu);
  HIP_TRY(hipSetDevice(h->device));
  return noise_floor_run(h, d_features, n_epochs, bands_of(h), nf_out, static_cast<hipStream_t>(stream));
}

int crn_sense_reserve_noise_floor(crn_handle *h) {
  if (!h) return crn::fail(CRN_ERR_ARG, "null handle");
  std::lock_guard<std::mutex> nf(h->nf_mu);
  HIP_TRY(hipSetDevice(h->device));
  const size_t bytes = (size_t)crn::kNoiseFloorMaxEpochs * CRN_MAX_BANDS * sizeof(float);   // any band plan the handle may get later
  if (!h->d_nf_scratch) HIP_TRY(hipMalloc(&h->d_nf_scratch, (crn::kNoiseFloorMaxEpochs + 1) * sizeof(float)));
  if (!h->h_nf_features) HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&h->h_nf_features), bytes, hipHostMallocDefault));
  if (!h->d_nf_features) HIP_TRY(hipMalloc(reinterpret_cast<void **>(&h->d_nf_features), bytes));
  return CRN_OK;
}

int crn_sense_set_thresholds(crn_handle *h, const float *thresh, int32_t n_bands, void *stream) {
  if (!h || !thresh) return crn::fail(CRN_ERR_ARG, "null handle / thresholds");
  if (int rc = refuse_capture(static_cast<hipStream_t>(stream), "crn_sense_set_thresholds")) return rc;
  std::unique_lock<std::mutex> lk(h->tables_mu);
  if (n_bands != h->cfg.n_bands) return crn::fail(CRN_ERR_ARG, "n_bands differs from the handle's");
  HIP_TRY(hipSetDevice(h->device));
  return set_thresholds_locked(h, lk, thresh, n_bands, static_cast<hipStream_t>(stream));
}

int crn_sense_calibrate_thresholds(crn_handle *h, const float *features, int64_t n_epochs, float lambda, float *nf_out, void *stream) {
  if (!h || !features || !nf_out) return crn::fail(CRN_ERR_ARG, "null handle / features / result");
  if (n_epochs < 1 || n_epochs > crn::kNoiseFloorMaxEpochs) return crn::fail(CRN_ERR_ARG, "n_epochs must be in 1..4096");
  if (!(lambda > 0.f)) return crn::fail(CRN_ERR_ARG, "lambda must be positive");
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (int rc = refuse_capture(st, "crn_sense_calibrate_thresholds")) return rc;
  int n_bands = 0;
  {
    // upload + reduction + the wait for its result under the noise-floor buffers' own lock: launches on other threads (an ingest ring's
    // launcher calls this between batches; the owner of the handle may be launching) are not held up by a stream drain
    std::lock_guard<std::mutex> nf(h->nf_mu);
    if (!h->h_nf_features || !h->d_nf_features || !h->d_nf_scratch)
      return crn::fail(CRN_ERR_STATE, "crn_sense_calibrate_thresholds: call crn_sense_reserve_noise_floor first (this call allocates nothing)");
    HIP_TRY(hipSetDevice(h->device));
    n_bands = bands_of(h);
    const size_t bytes = (size_t)n_epochs * n_bands * sizeof(float);
    std::memcpy(h->h_nf_features, features, bytes);
    HIP_TRY(hipMemcpyAsync(h->d_nf_features, h->h_nf_features, bytes, hipMemcpyHostToDevice, st));
    if (int rc = noise_floor_run(h, h->d_nf_features, n_epochs, n_bands, nf_out, st)) return rc;
  }
  float thr[CRN_MAX_BANDS];
  for (int b = 0; b < n_bands; b++) thr[b] = lambda * *nf_out;
  std::unique_lock<std::mutex> lk(h->tables_mu);
  if (n_bands != h->cfg.n_bands) return crn::fail(CRN_ERR_STATE, "crn_sense_calibrate_thresholds: the band plan changed while the noise floor was being estimated");
  return set_thresholds_locked(h, lk, thr, n_bands, st);
}

int crn_sense_set_ann(crn_handle *h, const double w_ih[5][6], const double w_ho[6][4], double threshold, void *stream) {
  if (!h || !w_ih || !w_ho) return crn::fail(CRN_ERR_ARG, "null handle / weights");
  if (int rc = refuse_capture(static_cast<hipStream_t>(stream), "crn_sense_set_ann")) return rc;
  std::unique_lock<std::mutex> lk(h->tables_mu);
  if (h->cfg.decide != CRN_DECIDE_ANN) return crn::fail(CRN_ERR_STATE, "crn_sense_set_ann: the handle does not decide with the network");
  if (!(threshold > 0.0 && threshold < 1.0)) return crn::fail(CRN_ERR_ARG, "threshold must be in (0, 1)");
  for (int i = 0; i < 5; i++)
    for (int j = 0; j < 6; j++)
      if (!std::isfinite(w_ih[i][j])) return crn::fail(CRN_ERR_ARG, "non-finite weight");
  for (int j = 0; j < 6; j++)
    for (int k = 0; k < 4; k++)
      if (!std::isfinite(w_ho[j][k])) return crn::fail(CRN_ERR_ARG, "non-finite weight");
  HIP_TRY(hipSetDevice(h->device));
  int slot = 0;
  bool waited = false;
  if (int rc = take_update_slot(h, lk, &slot, &waited)) return rc;
  if (waited && h->cfg.decide != CRN_DECIDE_ANN)
    return crn::fail(CRN_ERR_STATE, "crn_sense_set_ann: the handle's plan changed while this update waited for a staging slot: nothing was written");
  std::memcpy(h->cfg.ann_w_ih, w_ih, sizeof(h->cfg.ann_w_ih));
  std::memcpy(h->cfg.ann_w_ho, w_ho, sizeof(h->cfg.ann_w_ho));
  h->cfg.ann_threshold = threshold;   // rides in the launch parameters
  crn_handle::UpdateSlot &u = h->upd[slot];
  std::memcpy(u.w_ih, w_ih, sizeof(u.w_ih));
  std::memcpy(u.w_ho, w_ho, sizeof(u.w_ho));
  // the device copies the kernels read: the two tables and the packed band table's weight words (layout: crn_kernels.h)
  hipStream_t st = static_cast<hipStream_t>(stream);
  HIP_TRY(hipMemcpyAsync(const_cast<double *>(h->d_wih), u.w_ih, sizeof(u.w_ih), hipMemcpyHostToDevice, st));
  HIP_TRY(hipMemcpyAsync(const_cast<double *>(h->d_who), u.w_ho, sizeof(u.w_ho), hipMemcpyHostToDevice, st));
  HIP_TRY(hipMemcpyAsync(const_cast<int *>(h->d_band_tab) + 544, u.w_ih, sizeof(u.w_ih), hipMemcpyHostToDevice, st));
  HIP_TRY(hipMemcpyAsync(const_cast<int *>(h->d_band_tab) + 604, u.w_ho, sizeof(u.w_ho), hipMemcpyHostToDevice, st));
  HIP_TRY(hipEventRecord(h->upd_done[slot], st));
  return CRN_OK;
}

int crn_sense_set_bands(crn_handle *h, const crn_band_seg *segs, int32_t n_segs, int32_t n_bands, const float *thresh) {
  if (!h || !segs) return crn::fail(CRN_ERR_ARG, "null handle / segments");
  if (n_segs < 1 || n_segs > CRN_MAX_SEGS) return crn::fail(CRN_ERR_ARG, "n_segs out of range");
  // held across the rebuild AND the release of the old slab: a launch on another thread (an ingest ring's launcher) either was
  // enqueued before — hipFree inside build_tables waits for it — or starts after, with the new plan, whole
  std::lock_guard<std::mutex> lk(h->tables_mu);
  crn_cfg next = h->cfg;
  next.n_segs = n_segs;
  next.n_bands = n_bands;
  std::memcpy(next.segs, segs, sizeof(crn_band_seg) * (size_t)n_segs);
  if (thresh) {
    if (n_bands >= 1 && n_bands <= CRN_MAX_BANDS) std::memcpy(next.thresh, thresh, sizeof(float) * (size_t)n_bands);
  } else if (n_bands != h->cfg.n_bands) {
    return crn::fail(CRN_ERR_ARG, "crn_sense_set_bands: a different number of bands needs its thresholds");
  }
  if (int rc = validate(&next)) return rc;   // same rules as crn_sense_create (DECIDE_ANN keeps its 4 bands, ref_band stays inside)
  if (n_bands != h->cfg.n_bands && h->n_rings.load(std::memory_order_acquire) > 0)
    return crn::fail(CRN_ERR_STATE, "crn_sense_set_bands: an ingest ring on this handle was sized for the current number of bands "
                                    "(destroy it, change the plan, create it again)");
  HIP_TRY(hipSetDevice(h->device));
  const crn_cfg prev = h->cfg;
  h->cfg = next;
  if (int rc = build_tables(h)) {            // a fresh slab; the old one is freed once the device is idle
    h->cfg = prev;
    return rc;
  }
  return CRN_OK;
}

int crn_sense_synchronize(crn_handle *h, void *stream) {
  if (!h) return crn::fail(CRN_ERR_ARG, "null handle");
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
  return CRN_OK;
}

int crn_noise_floor_host(crn_handle *h, const float *features, int64_t n_epochs, float *nf_out) {
  if (!h || !features || !nf_out) return crn::fail(CRN_ERR_ARG, "null handle / features / result");
  if (n_epochs < 1) return crn::fail(CRN_ERR_ARG, "n_epochs < 1");
  if (int rc = crn_sense_reserve_noise_floor(h)) return rc;   // allocates on the first call only
  std::lock_guard<std::mutex> nf(h->nf_mu);
  HIP_TRY(hipSetDevice(h->device));
  const int n_bands = bands_of(h);
  const int64_t n = std::min<int64_t>(n_epochs, crn::kNoiseFloorMaxEpochs);
  const size_t bytes = (size_t)n * n_bands * sizeof(float);
  std::memcpy(h->h_nf_features, features, bytes);
  HIP_TRY(hipMemcpyAsync(h->d_nf_features, h->h_nf_features, bytes, hipMemcpyHostToDevice, nullptr));
  return noise_floor_run(h, h->d_nf_features, n, n_bands, nf_out, nullptr);
}

int crn_monitor_rows_device(crn_handle *h, const float *d_spectrum, int64_t n_rows, int32_t kind, float alpha,
                            int32_t first, float *d_state, float *d_waterfall_db, float *d_average_db, void *stream) {
  if (!h || !d_spectrum || !d_state) return crn::fail(CRN_ERR_ARG, "null handle / spectrum / state");
  if (n_rows < 0) return crn::fail(CRN_ERR_ARG, "n_rows < 0");
  if (kind != CRN_MONITOR_GNURADIO && kind != CRN_MONITOR_PSD) return crn::fail(CRN_ERR_ARG, "unknown monitor kind");
  if (!(alpha > 0.f && alpha <= 1.f)) return crn::fail(CRN_ERR_ARG, "alpha must be in (0, 1]");
  std::lock_guard<std::mutex> lk(h->tables_mu);
  HIP_TRY(hipSetDevice(h->device));
  const double N = (double)h->cfg.fft_len;
  crn::MonitorParams p{};
  p.spectrum = d_spectrum;
  p.n_rows = n_rows;
  p.n = h->cfg.fft_len;
  p.alpha = alpha;
  p.scale = (float)(1.0 / (kind == CRN_MONITOR_GNURADIO ? N * N : N * h->window_power));
  p.db_domain = kind == CRN_MONITOR_GNURADIO;
  p.first = first != 0;
  p.state = d_state;
  p.waterfall_db = d_waterfall_db;
  p.average_db = d_average_db;
  HIP_TRY(crn::launch_monitor(p, static_cast<hipStream_t>(stream)));
  return CRN_OK;
}

int crn_fft_forward_device(crn_handle *h, const float *d_in, int64_t n_frames, int32_t samples_per_frame,
                           int64_t frame_stride, float *d_out, void *stream) {
  if (!h || !d_in || !d_out) return crn::fail(CRN_ERR_ARG, "null handle / buffer");
  if (n_frames < 0) return crn::fail(CRN_ERR_ARG, "negative frame count");
  std::lock_guard<std::mutex> lk(h->tables_mu);
  if (samples_per_frame < 1 || samples_per_frame > h->cfg.fft_len)
    return crn::fail(CRN_ERR_ARG, "samples_per_frame must be in 1..fft_len");
  if (frame_stride <= 0) frame_stride = samples_per_frame;
  HIP_TRY(hipSetDevice(h->device));
  crn::FftParams p{};
  p.in = reinterpret_cast<const float2 *>(d_in);
  p.out = reinterpret_cast<float2 *>(d_out);
  p.n_frames = n_frames;
  p.frame_stride = frame_stride;
  p.L = samples_per_frame;
  p.tw1 = h->d_tw1;
  p.tw2 = h->d_tw2;
  HIP_TRY(crn::launch_fft(p, h->cfg.fft_len, static_cast<hipStream_t>(stream)));
  return CRN_OK;
}

int crn_synth_fill_device(crn_handle *h, float *d_iq, int64_t n_epochs, int64_t samples_per_epoch,
                          uint64_t seed, float noise_power, float signal_rms, int32_t tones_per_band,
                          int32_t *d_truth, void *stream) {
  crn_synth_cfg sc{};
  sc.seed = seed;
  sc.noise_power = noise_power;
  sc.signal_rms = signal_rms;
  sc.tones_per_band = tones_per_band;
  sc.pu_model = CRN_PU_UNIFORM;
  sc.signal_kind = CRN_SIG_TONES;
  sc.n_streams = 1;
  sc.adc_bits = 0;
  return crn_synth_fill_device_ex(h, &sc, d_iq, n_epochs, samples_per_epoch, d_truth, stream);
}

int crn_synth_fill_device_ex(crn_handle *h, const crn_synth_cfg *sc, float *d_iq, int64_t n_epochs,
                             int64_t samples_per_epoch, int32_t *d_truth, void *stream) {
  if (!h || !d_iq || !sc) return crn::fail(CRN_ERR_ARG, "null handle / configuration / IQ pointer");
  if (n_epochs < 0 || samples_per_epoch < 1) return crn::fail(CRN_ERR_ARG, "bad sizes");
  if (sc->tones_per_band < 0 || sc->noise_power < 0.f) return crn::fail(CRN_ERR_ARG, "bad signal parameters");
  if (sc->pu_model < CRN_PU_UNIFORM || sc->pu_model > CRN_PU_SWEEP)
    return crn::fail(CRN_ERR_ARG, "unknown pu_model");
  if (sc->signal_kind < CRN_SIG_TONES || sc->signal_kind > CRN_SIG_OFDM)
    return crn::fail(CRN_ERR_ARG, "unknown signal_kind");
  if (sc->n_streams < 1) return crn::fail(CRN_ERR_ARG, "n_streams must be >= 1");
  if (sc->adc_bits != 0 && (sc->adc_bits < 2 || sc->adc_bits > 24)) return crn::fail(CRN_ERR_ARG, "adc_bits must be 0 or 2..24");
  const bool markov = sc->pu_model == CRN_PU_MARKOV_AS_WRITTEN || sc->pu_model == CRN_PU_MARKOV_INTENDED;
  if (sc->pu_model != CRN_PU_UNIFORM) {
    if (markov && !d_truth) return crn::fail(CRN_ERR_ARG, "the Markov traffic models need d_truth");
    if (n_epochs % sc->n_streams != 0) return crn::fail(CRN_ERR_ARG, "n_streams must divide n_epochs");
  }
  std::lock_guard<std::mutex> lk(h->tables_mu);
  const crn_cfg &c = h->cfg;
  HIP_TRY(hipSetDevice(h->device));
  crn::SynthParams p{};
  p.iq = reinterpret_cast<float2 *>(d_iq);
  p.n_epochs = n_epochs;
  p.samples_per_epoch = samples_per_epoch;
  p.seed = sc->seed;
  p.noise_sigma = std::sqrt(sc->noise_power * 0.5f);
  p.tones = sc->tones_per_band;
  p.tone_amp = sc->tones_per_band > 0 ? sc->signal_rms / std::sqrt((float)sc->tones_per_band) : 0.f;
  p.signal_rms = sc->signal_rms;
  p.pu_model = sc->pu_model;
  p.signal_kind = sc->signal_kind;
  p.epochs_per_stream = n_epochs / sc->n_streams;
  p.adc_scale = sc->adc_bits ? (float)(1 << (sc->adc_bits - 1)) : 0.f;
  p.fft_len = c.fft_len;
  if (c.ref_band >= 0 || c.decide == CRN_DECIDE_ANN) {  // {NF, CH1, ..}: band 0 is never driven
    p.active_band0 = 1;
    p.n_active = std::min(3, c.n_bands - 1);
  } else {
    p.active_band0 = 0;
    p.n_active = c.n_bands;
  }
  if (sc->signal_kind == CRN_SIG_TONES && sc->tones_per_band == 0) p.n_active = 0;
  p.band_bins_begin = h->d_band_bins_begin;
  p.band_bins = h->d_band_bins;
  p.band_c2 = h->d_band_c2;
  p.truth = d_truth;
  if (sc->pu_model != CRN_PU_UNIFORM && p.n_active < 1)
    return crn::fail(CRN_ERR_ARG, "the Markov and sweep traffic models need at least one driven band");
  if (markov) HIP_TRY(crn::launch_pu_pattern(p, static_cast<hipStream_t>(stream)));
  HIP_TRY(crn::launch_synth(p, static_cast<hipStream_t>(stream)));
  return CRN_OK;
}

}  // extern "C"
