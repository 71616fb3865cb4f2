// crn_ingest.cpp — packet ingest ring (include/crn_sense.h, "ingest ring").
//
// Host-side counterpart of the ECR's rx -> CE hand-off
// (reference: src/extensible_cognitive_radio.cpp:1310-1324).  Two pinned batch buffers of C epoch
// slots each.  A stream that starts an epoch is given the next free slot of the batch being filled
// and every packet is copied ONCE, straight into its place in that pinned slot; when enough slots
// are complete the batch is handed to the ring's launcher thread, which makes every HIP call
// (H2D -> sensing kernel -> D2H on a private stream, completion by event) and files the results;
// filling continues in the other buffer.
//
// Disjoint frames (hop == fft_len: the reference's own shape): a packet is a frame, an epoch is K packets, the kernel zero-pads
// each to N.  Overlapped frames (hop < fft_len: the Welch plans): the packets of an epoch are laid end to end as one contiguous
// run of samples — P = ceil(((K - 1) hop + N) / L) of them — and the kernel cuts its K overlapped N-sample frames out of that run
// (epoch stride P L).
//
// Threads: the caller's thread (the CE thread: crn_ingest_push / _poll run inside execute(), with
// CE_mutex held) touches pinned memory, a few counters, and a mutex for the hand-off queues — never
// the HIP runtime, never the allocator on the packet path.  When the buffer it would have to fill is
// still on the GPU the packet is refused with CRN_ERR_BUSY (the reference's own hand-off drops
// frames the CE thread is not ready for).
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/crn_sense.h"
#ifdef CRN_WITH_SC16
#include "../../include/crn_sense_sc16.h"   // the optional wire-format entry points (libcrnsense_sc16.so)
#endif
#include "crn_internal.h"

#define HIP_TRY(expr)                                                                      \
  do {                                                                                     \
    hipError_t _e = (expr);                                                                \
    if (_e != hipSuccess)                                                                  \
      return crn::fail(CRN_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(_e)); \
  } while (0)

namespace {

struct Slot {
  int32_t stream;  // -1: a hole (its epoch moved on to the next batch; nothing to report)
  int64_t seq;
  int npk;         // packets staged (of P)
};

enum : int { kFree = 0, kQueued = 1 };  // Batch::state: owned by the caller / by the launcher thread

struct Batch {
  char *h_iq = nullptr;         // pinned [C][K][cap] samples (complex floats, or int16 pairs in a wire-format ring)
  char *d_iq = nullptr;
  char *h_res = nullptr;        // pinned results: features | ann | decision | occupancy
  char *d_res = nullptr;
  hipEvent_t done = nullptr;
  std::vector<Slot> slots;      // [C]
  int assigned = 0;             // slots handed to streams, in order of epoch start
  int complete = 0;             // of which hold all K packets
  int staged = 0;               // packets staged in this buffer (with the packets of epochs carried over from the other one)
  int launched = 0;             // slots of the launch handed to the launcher thread
  int L = 0, P = 0;             // packet length and packets per epoch of that launch
  bool calibrating = false;     // launched while a calibration was collecting or before its thresholds were in place (launcher thread)
  std::chrono::steady_clock::time_point t_handoff;   // when the pushing thread handed it over (written before state = kQueued)
  std::chrono::steady_clock::time_point t_dequeued, t_enqueued;   // launcher thread ($CRN_INGEST_TRACE)
  std::atomic<int> state{kFree};
};

}  // namespace

struct crn_ingest {
  crn_handle *h = nullptr;
  crn_cfg cfg;
  int n_streams = 0, L = 0, cap = 0, B = 0, C = 0, K = 0;
  int P = 0;                               // packets per epoch: K (disjoint frames) or ceil(((K - 1) hop + N) / L) (overlapped)
  bool overlapped = false;                 // cfg.hop < cfg.fft_len
  size_t sample_bytes = 8;                 // 8: complex floats; 4: the radio's wire format (crn_ingest_create_sc16)
  size_t epoch_bytes = 0;                  // P * L * sample_bytes: slot pitch for the current packet length
  size_t epoch_cap_bytes = 0;              // largest epoch_bytes any packet length <= cap can need (buffers are sized for it)
  size_t off_ann = 0, off_dec = 0, off_occ = 0, res_bytes = 0;
  hipStream_t stream = nullptr;
  Batch batch[2];
  // ---- caller-thread state ----
  int fill = 0;                            // batch being filled; every open slot lives in it
  std::vector<int> open_slot;              // per stream: its slot in batch[fill], -1 = between epochs
  std::vector<int64_t> seq;                // per stream: epochs started
  int64_t dropped = 0;                     // packets refused with CRN_ERR_BUSY
  int64_t packets = 0;                     // packets accepted
  int64_t polled = 0;                      // results handed out
  // ---- shared with the launcher thread (under mu) ----
  std::mutex mu;
  std::condition_variable cv_work, cv_free;
  std::deque<int> work;                    // batches to launch, in order
  std::deque<crn_epoch_result> ready;
  // crn_ingest_poll runs in every execute() — tens of thousands of times a second when the CE thread spins (ce_timeout_ms = 0) — and a
  // result arrives once per sensing period: the poll reads this flag first and takes `mu` only when there is something to fetch, so
  // that execute() (CE_mutex held) never queues behind a launcher thread the scheduler has put to sleep with `mu` in hand.
  std::atomic<bool> poll_hint{false};      // ready is not empty, or err_code is set (written under mu)
  int err_code = CRN_OK;                   // first failure on the launcher thread, reported by the next call
  std::string err_msg;
  bool stop = false;
  size_t zero_copy_bytes = 512 * 1024;     // batches up to this size are read / written in place by the kernel
  int spin_us = 150;                       // how long after a hand-off the launcher keeps polling before it sleeps ($CRN_INGEST_SPIN_US)
  std::atomic<bool> work_waiting{false};   // work is not empty (read by the launcher without the lock while it polls an event)
  // Pre-wake: a sleeping launcher thread needs tens of microseconds to come back (63 us from hand-off to decision with the GPU and
  // the thread idle for 100 ms between epochs, against 37 us back to back: tools/engine_idle_gap.py).  With a small batch — the
  // engine's one epoch — the first packet of a batch tells the launcher that a hand-off is K packet times away (280 us at 13 Msps):
  // it wakes up now and polls for the work instead of sleeping, for at most prewake_us ($CRN_INGEST_PREWAKE_US, 0 = never).
  std::atomic<bool> prewake{false};
  int prewake_us = 600;
  bool warm_gpu = true;                    // $CRN_INGEST_WARM_GPU=0: no empty launch at the pre-wake
  std::chrono::steady_clock::time_point t_last_launch{};   // launcher thread: when it last enqueued a batch
  int prewake_at = 0;                      // packets staged in a batch at which the launcher is told (0 = never): B P - 10, at least 1 — ten
                                           // packets before the hand-off (280 us at 13 Msps), only for batches small enough for the in-place launch
  // $CRN_INGEST_TRACE=1: where the hand-off-to-results time goes, summed on the launcher thread and printed by crn_ingest_destroy:
  // hand-off -> launcher has the batch (its wake-up), -> launch enqueued (the HIP calls), -> results seen (device time + noticing)
  bool trace = false;
  double tr_wake_us = 0, tr_enqueue_us = 0, tr_device_us = 0;
  int64_t tr_n = 0;
  int64_t n_batches = 0, n_failed = 0, n_epochs_launched = 0, n_epochs_ready = 0;   // crn_ingest_get_stats
  double lat_us_sum = 0.0, lat_us_max = 0.0;
  // ---- noise-floor calibration (crn_ingest_calibrate): requested by the caller's thread, carried out by the launcher ----
  std::atomic<int> calib_state{0};         // 0 none / done, 1 collecting (set under mu by the caller, cleared by the launcher)
  int calib_target = 0, calib_have = 0;    // epochs wanted / collected (launcher thread once calib_state == 1)
  float calib_lambda = 0.f;
  std::vector<float> calib_feat;           // [kCalibMaxEpochs][n_bands], sized at creation
  std::atomic<float> noise_floor{0.f};     // the estimate in force (0 until a calibration has finished)
  std::thread launcher;
  bool attached = false;                   // counted on the handle (crn_sense_set_bands refuses to change n_bands under a ring)
};

constexpr int kCalibMaxEpochs = 4096;      // crn_noise_floor_device uses at most this many

// defined in crn_api.cpp
extern "C" int crn_sense_cfg_of(crn_handle *h, crn_cfg *out);
extern "C" int crn_sense_ring_count(crn_handle *h, int delta);
extern "C" int crn_sense_warm_stream(crn_handle *h, void *stream);
extern "C" int crn_sense_run_device_any(crn_handle *h, const void *d_iq, int32_t bytes_per_sample, int64_t n_epochs, int32_t samples_per_frame,
                                        int64_t epoch_stride, const crn_out *d_out, void *stream);

namespace {

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// packets of L samples that make up one epoch
int packets_per_epoch(const crn_cfg &c, int L) {
  if (c.hop == c.fft_len) return c.frames_per_epoch;
  const long long span = (long long)(c.frames_per_epoch - 1) * c.hop + c.fft_len;
  return (int)((span + L - 1) / L);
}

inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
  __builtin_ia32_pause();
#else
  std::this_thread::yield();
#endif
}

void set_prewake(crn_ingest *g) {
  const long long total = (long long)g->B * g->P;
  const bool small = (size_t)g->B * g->epoch_bytes <= g->zero_copy_bytes;
  g->prewake_at = (small && total > 1 && g->prewake_us > 0) ? (int)std::max<long long>(total - 10, 1) : 0;
}

// ---- launcher thread -------------------------------------------------------------------------

// H2D + kernel + D2H + event for one batch; returns an error message or "".
std::string enqueue(crn_ingest *g, Batch &b) {
  const size_t in_bytes = (size_t)b.launched * (size_t)b.P * b.L * g->sample_bytes;
  // a small batch (the engine's one epoch): the kernel reads the pinned slots and writes the pinned results over the bus itself —
  // one launch instead of upload + launch + download: the kernel takes 30 instead of 23 us for one reference epoch, the results
  // are readable 15-19 us sooner (tools/ring_rate: 89 -> 74 us; 8 epochs: 92 -> 73 us).  $CRN_INGEST_ZEROCOPY_BYTES: largest
  // batch handled so (0 = never)
  const bool zero_copy = in_bytes <= g->zero_copy_bytes;
  // decided against whatever thresholds are in place now: while a calibration is collecting those are not the ones asked for
  b.calibrating = g->calib_state.load(std::memory_order_acquire) != 0;
  hipError_t e = hipSuccess;
  if (!zero_copy) e = hipMemcpyAsync(b.d_iq, b.h_iq, in_bytes, hipMemcpyHostToDevice, g->stream);
  if (e != hipSuccess) return std::string("hipMemcpyAsync(H2D): ") + hipGetErrorString(e);
  char *res = zero_copy ? b.h_res : b.d_res;
  crn_out out;
  out.features = reinterpret_cast<float *>(res);
  out.ann_out = reinterpret_cast<double *>(res + g->off_ann);
  out.decision = reinterpret_cast<int32_t *>(res + g->off_dec);
  out.occupancy = reinterpret_cast<uint8_t *>(res + g->off_occ);
  out.spectrum = nullptr;
  const void *src = zero_copy ? b.h_iq : b.d_iq;
  // disjoint frames: K packets zero-padded to N each, dense epochs; overlapped: whole frames cut from the epoch's run of P L samples
  const int32_t spf = g->overlapped ? g->cfg.fft_len : b.L;
  const int64_t stride = g->overlapped ? (int64_t)b.P * b.L : 0;
  const int rc = crn_sense_run_device_any(g->h, src, (int32_t)g->sample_bytes, b.launched, spf, stride, &out, g->stream);
  if (rc != CRN_OK) return crn_last_error();
  if (!zero_copy) e = hipMemcpyAsync(b.h_res, b.d_res, g->res_bytes, hipMemcpyDeviceToHost, g->stream);
  if (e == hipSuccess) e = hipEventRecord(b.done, g->stream);
  if (e != hipSuccess) return std::string("hipMemcpyAsync(D2H) / hipEventRecord: ") + hipGetErrorString(e);
  return "";
}

void collect(crn_ingest *g, const Batch &b, std::vector<crn_epoch_result> *out) {
  const int nb = g->cfg.n_bands;
  const float *feat = reinterpret_cast<const float *>(b.h_res);
  const double *ann = reinterpret_cast<const double *>(b.h_res + g->off_ann);
  const int32_t *dec = reinterpret_cast<const int32_t *>(b.h_res + g->off_dec);
  const uint8_t *occ = reinterpret_cast<const uint8_t *>(b.h_res + g->off_occ);
  for (int s = 0; s < b.launched; s++) {
    if (b.slots[s].stream < 0) continue;
    crn_epoch_result r;
    std::memset(&r, 0, sizeof(r));
    r.stream = b.slots[s].stream;
    r.epoch_seq = b.slots[s].seq;
    r.decision = dec[s];
    if (g->cfg.decide == CRN_DECIDE_ANN) std::memcpy(r.ann_out, ann + 3 * s, 3 * sizeof(double));
    std::memcpy(r.features, feat + (size_t)s * nb, nb * sizeof(float));
    std::memcpy(r.occupancy, occ + (size_t)s * nb, nb);
    r.flags = b.calibrating ? CRN_EPOCH_CALIBRATION : 0;
    r.noise_floor = b.calibrating ? 0.f : g->noise_floor.load(std::memory_order_relaxed);
    out->push_back(r);
  }
}

// Launcher thread, after a batch has come back: feed a calibration that is collecting; with the last epoch it wants, estimate the
// noise floor on the device and put lambda x the estimate in as every band's threshold, ordered on the ring's stream — batches
// launched from here on are decided against it (those already in flight keep their CRN_EPOCH_CALIBRATION mark).  Blocks this
// thread for one small upload + reduction; never the pushing thread.  Returns an error message or "".
std::string feed_calibration(crn_ingest *g, const std::vector<crn_epoch_result> &res) {
  if (g->calib_state.load(std::memory_order_acquire) == 0) return "";
  const int nb = g->cfg.n_bands;
  for (const crn_epoch_result &r : res) {
    if (g->calib_have >= g->calib_target) break;
    std::memcpy(&g->calib_feat[(size_t)g->calib_have++ * nb], r.features, sizeof(float) * (size_t)nb);
  }
  if (g->calib_have < g->calib_target) return "";
  float nf = 0.f;
  const int rc = crn_sense_calibrate_thresholds(g->h, g->calib_feat.data(), g->calib_target, g->calib_lambda, &nf, g->stream);
  g->calib_state.store(0, std::memory_order_release);   // also after a failure: the run goes on with the thresholds it had
  if (rc != CRN_OK) return std::string("noise-floor calibration failed: ") + crn_last_error();
  g->noise_floor.store(nf, std::memory_order_release);
  return "";
}

// Give a batch back to the caller's thread (mu held).
void release_batch(crn_ingest *g, Batch &b) {
  b.assigned = b.complete = b.launched = b.staged = 0;
  b.state.store(kFree, std::memory_order_release);
  g->cv_free.notify_all();
}

void launcher_main(crn_ingest *g) {
  (void)hipSetDevice(g->cfg.device);
  std::deque<int> inflight;
  std::vector<crn_epoch_result> res;
  std::unique_lock<std::mutex> lk(g->mu);
  for (;;) {
    while (!g->work.empty()) {   // launch everything that is queued before waiting for anything
      const int i = g->work.front();
      g->work.pop_front();
      if (g->work.empty()) g->work_waiting.store(false, std::memory_order_release);
      g->prewake.store(false, std::memory_order_release);   // the hand-off it announced is here
      lk.unlock();
      if (g->trace) g->batch[i].t_dequeued = std::chrono::steady_clock::now();
      const std::string err = enqueue(g, g->batch[i]);
      g->t_last_launch = std::chrono::steady_clock::now();
      if (g->trace) g->batch[i].t_enqueued = g->t_last_launch;
      lk.lock();
      if (!err.empty()) {
        if (g->err_code == CRN_OK) {
          g->err_code = CRN_ERR_DEVICE;
          g->err_msg = "ingest launch failed, batch dropped: " + err;
          g->poll_hint.store(true, std::memory_order_release);
        }
        g->n_failed++;
        release_batch(g, g->batch[i]);
      } else {
        g->n_batches++;
        for (int sl = 0; sl < g->batch[i].launched; sl++) g->n_epochs_launched += g->batch[i].slots[sl].stream >= 0;   // holes carry nothing
        inflight.push_back(i);
      }
    }
    if (inflight.empty()) {
      if (g->stop) return;
      if (g->prewake.exchange(false, std::memory_order_acq_rel)) {
        // a batch has started to fill: stay awake for its hand-off (bounded; the pushing thread never waits for this)
        lk.unlock();
        const auto t0 = std::chrono::steady_clock::now();
        const auto budget = std::chrono::microseconds(g->prewake_us);
        // ... and the queue with it when it has sat idle: an empty launch now, well ahead of the one that matters
        if (g->warm_gpu && t0 - g->t_last_launch > std::chrono::milliseconds(2)) (void)crn_sense_warm_stream(g->h, g->stream);
        while (!g->work_waiting.load(std::memory_order_acquire) && std::chrono::steady_clock::now() - t0 < budget) cpu_relax();
        lk.lock();
        continue;
      }
      g->cv_work.wait(lk);
      continue;
    }
    Batch &b = g->batch[inflight.front()];
    lk.unlock();
    // a small batch (the engine's shape: one epoch) is back within ~50 us of its launch and a sleeping thread wakes up tens of
    // microseconds late: keep looking for up to 150 us after the hand-off unless new work is waiting to be launched (hand-off to
    // decision 106 -> 58 us).  Big batches take as long to fill as to run and polling through them costs the pushing thread
    // 10-30 % of its rate (measured, tools/ring_rate): they sleep.
    const bool small = (size_t)b.launched * (size_t)b.P * (size_t)b.L * g->sample_bytes <= g->zero_copy_bytes;
    const auto spin = std::chrono::microseconds(small ? g->spin_us : 0);
    hipError_t q = hipEventQuery(b.done);
    while (q == hipErrorNotReady && !g->work_waiting.load(std::memory_order_acquire) && std::chrono::steady_clock::now() - b.t_handoff < spin)
      q = hipEventQuery(b.done);
    if (q == hipSuccess) {
      res.clear();
      collect(g, b, &res);
      const std::string cal_err = feed_calibration(g, res);
      const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - b.t_handoff).count();
      lk.lock();
      for (const crn_epoch_result &r : res) g->ready.push_back(r);
      g->n_epochs_ready += (int64_t)res.size();
      if (!cal_err.empty() && g->err_code == CRN_OK) {
        g->err_code = CRN_ERR_DEVICE;
        g->err_msg = cal_err;
      }
      if (!g->ready.empty() || g->err_code != CRN_OK) g->poll_hint.store(true, std::memory_order_release);
      g->lat_us_sum += us;
      if (us > g->lat_us_max) g->lat_us_max = us;
      if (g->trace) {
        const auto d = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b2) {
          return std::chrono::duration<double, std::micro>(b2 - a).count();
        };
        g->tr_wake_us += d(b.t_handoff, b.t_dequeued);
        g->tr_enqueue_us += d(b.t_dequeued, b.t_enqueued);
        g->tr_device_us += us - d(b.t_handoff, b.t_enqueued);
        g->tr_n++;
      }
      release_batch(g, b);
      inflight.pop_front();
      continue;
    }
    lk.lock();
    if (q != hipErrorNotReady) {
      if (g->err_code == CRN_OK) {
        g->err_code = CRN_ERR_DEVICE;
        g->err_msg = std::string("ingest batch failed on the device: ") + hipGetErrorString(q);
        g->poll_hint.store(true, std::memory_order_release);
      }
      g->n_failed++;
      release_batch(g, b);
      inflight.pop_front();
      continue;
    }
    // new work wakes it early.  (wait_until on the system clock = pthread_cond_timedwait, which ThreadSanitizer models;
    // wait_for would be pthread_cond_clockwait, which gcc 11's libtsan does not intercept: tests/harness/ring_unit.cpp)
    // A small batch is back within tens of microseconds: look again soon.  A big one takes as long to run as to fill: sleep
    // through most of it (200 us steps instead of 50 000 wake-ups a second).
    if (g->work.empty()) g->cv_work.wait_until(lk, std::chrono::system_clock::now() + std::chrono::microseconds(small ? 20 : 200));
  }
}

// ---- caller's thread --------------------------------------------------------------------------

bool is_free(const Batch &b) { return b.state.load(std::memory_order_acquire) == kFree; }

int sticky_error(crn_ingest *g) {
  std::lock_guard<std::mutex> lk(g->mu);
  if (g->err_code == CRN_OK) return CRN_OK;
  const int code = g->err_code;
  g->err_code = CRN_OK;
  g->poll_hint.store(!g->ready.empty(), std::memory_order_release);
  return crn::fail(code, g->err_msg);
}

// Hand the complete epochs of the batch being filled to the launcher thread (no HIP call, no wait).
// Epochs that are still open move on to the other buffer — which must be free for that: CRN_ERR_BUSY
// otherwise.
int launch(crn_ingest *g) {
  Batch &b = g->batch[g->fill];
  if (b.complete == 0) return CRN_OK;
  Batch &o = g->batch[g->fill ^ 1];
  const bool holes = b.complete != b.assigned;
  if (holes && !is_free(o)) return crn::fail(CRN_ERR_BUSY, "ingest ring: both batch buffers are in flight");
  const int n = b.assigned;
  b.launched = n;
  b.L = g->L;
  b.P = g->P;
  if (holes) {
    // open epochs continue in the other buffer (their packets so far are copied over: this only
    // happens when streams run at different rates or on an explicit flush)
    for (int i = 0; i < n; i++) {
      Slot &sl = b.slots[i];
      if (sl.stream < 0 || sl.npk == g->P) continue;
      const int j = o.assigned++;
      o.slots[j] = sl;
      o.staged += sl.npk;
      std::memcpy(o.h_iq + (size_t)j * g->epoch_bytes, b.h_iq + (size_t)i * g->epoch_bytes, (size_t)sl.npk * g->L * g->sample_bytes);
      g->open_slot[sl.stream] = j;
      sl.stream = -1;
    }
  }
  b.t_handoff = std::chrono::steady_clock::now();
  b.state.store(kQueued, std::memory_order_release);
  {
    std::lock_guard<std::mutex> lk(g->mu);
    g->work.push_back(g->fill);
    g->work_waiting.store(true, std::memory_order_release);
  }
  g->cv_work.notify_one();
  g->fill ^= 1;
  return CRN_OK;
}

bool launch_due(const crn_ingest *g, const Batch &b) {
  // once epochs_per_batch are complete — hole-free if possible (streams fed round-robin complete
  // together), with holes only when the buffer has no slot left
  return b.complete >= g->B && (b.complete == b.assigned || b.assigned == g->C);
}

void wait_free(crn_ingest *g, Batch &b) {
  std::unique_lock<std::mutex> lk(g->mu);
  g->cv_free.wait(lk, [&] { return is_free(b); });
}

}  // namespace

extern "C" {

static int ingest_create(crn_handle *h, int32_t n_streams, int32_t samples_per_packet, int32_t epochs_per_batch,
                         crn_ingest **out, size_t sample_bytes) {
  if (!h || !out) return crn::fail(CRN_ERR_ARG, "crn_ingest_create: null argument");
  *out = nullptr;
  crn_cfg cfg;
  if (int rc = crn_sense_cfg_of(h, &cfg)) return rc;
  if (n_streams < 1 || epochs_per_batch < 1) return crn::fail(CRN_ERR_ARG, "n_streams / epochs_per_batch < 1");
  if (samples_per_packet < 1 || samples_per_packet > cfg.fft_len)
    return crn::fail(CRN_ERR_ARG, "samples_per_packet must be in 1..fft_len");
  HIP_TRY(hipSetDevice(cfg.device));
  crn_ingest *g = new (std::nothrow) crn_ingest();
  if (!g) return crn::fail(CRN_ERR_NOMEM, "out of host memory");
  g->h = h;
  g->cfg = cfg;
  g->n_streams = n_streams;
  g->L = g->cap = samples_per_packet;
  g->B = epochs_per_batch;
  // every stream can have one epoch open: with C >= n_streams a full buffer always holds a complete one
  g->C = epochs_per_batch > n_streams ? epochs_per_batch : n_streams;
  g->K = cfg.frames_per_epoch;
  g->overlapped = cfg.hop != cfg.fft_len;
  g->P = packets_per_epoch(cfg, g->L);
  g->sample_bytes = sample_bytes;
  g->epoch_bytes = (size_t)g->P * g->L * sample_bytes;
  // overlapped: P(L) L < span + L for every L <= cap
  g->epoch_cap_bytes = g->overlapped ? ((size_t)(g->K - 1) * cfg.hop + cfg.fft_len + g->cap) * sample_bytes : g->epoch_bytes;
  const size_t nb = (size_t)cfg.n_bands;
  g->off_ann = align_up((size_t)g->C * nb * sizeof(float), 256);
  g->off_dec = g->off_ann + align_up((size_t)g->C * 3 * sizeof(double), 256);
  g->off_occ = g->off_dec + align_up((size_t)g->C * sizeof(int32_t), 256);
  g->res_bytes = g->off_occ + align_up((size_t)g->C * nb, 256);
  g->open_slot.assign(n_streams, -1);
  g->seq.assign(n_streams, 0);
  if (cfg.decide == CRN_DECIDE_THRESHOLD) {
    // everything a later crn_ingest_calibrate needs: the launcher's feature buffer, the handle's pinned + device upload buffers
    g->calib_feat.assign((size_t)kCalibMaxEpochs * nb, 0.f);
    if (int rc = crn_sense_reserve_noise_floor(h)) {
      crn_ingest_destroy(g);
      return rc;
    }
  }
  hipError_t e = hipStreamCreateWithFlags(&g->stream, hipStreamNonBlocking);
  for (int i = 0; i < 2 && e == hipSuccess; i++) {
    Batch &b = g->batch[i];
    b.slots.resize(g->C);
    const size_t iq_bytes = (size_t)g->C * g->epoch_cap_bytes;
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
  if (const char *e = std::getenv("CRN_INGEST_SPIN_US")) g->spin_us = std::max(0, std::atoi(e));
  if (const char *e = std::getenv("CRN_INGEST_ZEROCOPY_BYTES")) g->zero_copy_bytes = (size_t)std::max(0ll, std::atoll(e));
  if (const char *e = std::getenv("CRN_INGEST_PREWAKE_US")) g->prewake_us = std::min(100000, std::max(0, std::atoi(e)));
  set_prewake(g);
  if (const char *e = std::getenv("CRN_INGEST_TRACE")) g->trace = std::atoi(e) != 0;
  if (const char *e = std::getenv("CRN_INGEST_WARM_GPU")) g->warm_gpu = std::atoi(e) != 0;
  g->launcher = std::thread(launcher_main, g);
  (void)crn_sense_ring_count(h, +1);
  g->attached = true;
  *out = g;
  return CRN_OK;
}

int crn_ingest_calibrate(crn_ingest *g, int32_t n_epochs, float lambda) {
  if (!g) return crn::fail(CRN_ERR_ARG, "null ingest ring");
  if (g->cfg.decide != CRN_DECIDE_THRESHOLD) return crn::fail(CRN_ERR_STATE, "crn_ingest_calibrate: the handle does not decide by thresholds");
  if (n_epochs < 1 || n_epochs > kCalibMaxEpochs) return crn::fail(CRN_ERR_ARG, "n_epochs must be in 1..4096");
  if (!(lambda > 0.f)) return crn::fail(CRN_ERR_ARG, "lambda must be positive");
  std::lock_guard<std::mutex> lk(g->mu);
  if (g->calib_state.load(std::memory_order_acquire) != 0) return crn::fail(CRN_ERR_STATE, "crn_ingest_calibrate: a calibration is already collecting");
  g->calib_target = n_epochs;
  g->calib_have = 0;
  g->calib_lambda = lambda;
  g->calib_state.store(1, std::memory_order_release);   // the launcher reads the three fields above only after it sees this
  return CRN_OK;
}

int crn_ingest_noise_floor(crn_ingest *g, float *nf_out, int32_t *calibrating) {
  if (!g || !nf_out) return crn::fail(CRN_ERR_ARG, "null argument");
  *nf_out = g->noise_floor.load(std::memory_order_acquire);
  if (calibrating) *calibrating = g->calib_state.load(std::memory_order_acquire);
  return CRN_OK;
}

int crn_ingest_set_packet_len(crn_ingest *g, int32_t samples_per_packet) {
  if (!g) return crn::fail(CRN_ERR_ARG, "null ingest ring");
  if (samples_per_packet < 1 || samples_per_packet > g->cap)
    return crn::fail(CRN_ERR_ARG, "samples_per_packet must be in 1..the length the ring was created with");
  if (samples_per_packet == g->L) return CRN_OK;
  Batch &b = g->batch[g->fill];
  if (!is_free(b)) return crn::fail(CRN_ERR_BUSY, "ingest ring: both batch buffers are in flight");
  if (b.assigned != 0) return crn::fail(CRN_ERR_STATE, "crn_ingest_set_packet_len: epochs are staged (flush first)");
  g->L = samples_per_packet;
  g->P = packets_per_epoch(g->cfg, g->L);
  g->epoch_bytes = (size_t)g->P * g->L * g->sample_bytes;
  set_prewake(g);
  return CRN_OK;
}

static int ingest_push(crn_ingest *g, int32_t stream, const void *iq_packet, size_t sample_bytes) {
  if (!g || !iq_packet) return crn::fail(CRN_ERR_ARG, "crn_ingest_push: null argument");
  if (sample_bytes != g->sample_bytes)
    return crn::fail(CRN_ERR_STATE, g->sample_bytes == 4 ? "this ring takes wire-format packets: crn_ingest_push_sc16"
                                                          : "this ring takes complex-float packets: crn_ingest_push");
  if (stream < 0 || stream >= g->n_streams) return crn::fail(CRN_ERR_ARG, "stream id out of range");
  for (int attempt = 0;; attempt++) {
    Batch &b = g->batch[g->fill];
    if (!is_free(b)) {
      g->dropped++;
      return crn::fail(CRN_ERR_BUSY, "ingest ring: both batch buffers are in flight, packet refused");
    }
    int sl = g->open_slot[stream];
    if (sl < 0) {
      if (b.assigned == g->C) {  // no free slot: send the complete epochs off, carry the open ones over
        if (attempt > 0) return crn::fail(CRN_ERR_STATE, "ingest ring: no slot after a launch");
        if (int rc = launch(g)) {
          if (rc == CRN_ERR_BUSY) g->dropped++;
          return rc;
        }
        continue;  // the fill buffer changed
      }
      sl = b.assigned++;
      b.slots[sl] = Slot{stream, g->seq[stream]++, 0};
      g->open_slot[stream] = sl;
    }
    Slot &s = b.slots[sl];
    std::memcpy(b.h_iq + (size_t)sl * g->epoch_bytes + (size_t)s.npk * g->L * g->sample_bytes, iq_packet, (size_t)g->L * g->sample_bytes);
    g->packets++;
    // a small batch is prewake_packets away from its hand-off: have the launcher awake by then (see prewake)
    if (++b.staged == g->prewake_at && g->prewake_at > 0) {
      // No lock: execute() holds CE_mutex here and must not queue behind the launcher for a hint.  The flag is atomic; if the launcher
      // sits between its own test of the flag and its wait when this lands, the notification finds nobody and this batch's hand-off
      // wakes it the ordinary way — a missed pre-wake costs that one decision ~50 us, never a result.
      g->prewake.store(true, std::memory_order_release);
      g->cv_work.notify_one();
    }
    if (++s.npk < g->P) return CRN_OK;
    g->open_slot[stream] = -1;
    b.complete++;
    if (launch_due(g, b)) {
      const int rc = launch(g);
      return rc == CRN_ERR_BUSY ? CRN_OK : rc;  // this packet is staged; the hand-off is retried by the next push / poll
    }
    return CRN_OK;
  }
}

int crn_ingest_create(crn_handle *h, int32_t n_streams, int32_t samples_per_packet, int32_t epochs_per_batch, crn_ingest **out) {
  return ingest_create(h, n_streams, samples_per_packet, epochs_per_batch, out, 8);
}
int crn_ingest_push(crn_ingest *g, int32_t stream, const float *iq_packet) { return ingest_push(g, stream, iq_packet, 8); }
#ifdef CRN_WITH_SC16   // optional: rings of wire-format packets (make SC16=1 -> libcrnsense_sc16.so)
int crn_ingest_create_sc16(crn_handle *h, int32_t n_streams, int32_t samples_per_packet, int32_t epochs_per_batch, crn_ingest **out) {
  return ingest_create(h, n_streams, samples_per_packet, epochs_per_batch, out, 4);
}
int crn_ingest_push_sc16(crn_ingest *g, int32_t stream, const int16_t *iq_packet) { return ingest_push(g, stream, iq_packet, 4); }
#endif

int crn_ingest_flush(crn_ingest *g) {
  if (!g) return crn::fail(CRN_ERR_ARG, "null ingest ring");
  if (!is_free(g->batch[g->fill])) return CRN_OK;  // nothing can be staged in a buffer that is on the GPU
  int rc = launch(g);
  if (rc == CRN_ERR_BUSY) {  // open epochs need the other buffer: wait for it (flush may block, push never does)
    wait_free(g, g->batch[g->fill ^ 1]);
    rc = launch(g);
  }
  return rc;
}

int crn_ingest_wait(crn_ingest *g) {
  if (!g) return crn::fail(CRN_ERR_ARG, "null ingest ring");
  Batch &b = g->batch[g->fill];
  wait_free(g, b);
  // a push can also be refused because the hand-off of THIS buffer needs the other one (open epochs move over): then that is
  // the buffer to wait for, not the one being filled
  if (b.assigned == g->C && b.complete != b.assigned) wait_free(g, g->batch[g->fill ^ 1]);
  return sticky_error(g);
}

int crn_ingest_packets_per_epoch(crn_ingest *g, int32_t *n_packets) {
  if (!g || !n_packets) return crn::fail(CRN_ERR_ARG, "null argument");
  *n_packets = g->P;
  return CRN_OK;
}

int crn_ingest_poll(crn_ingest *g, crn_epoch_result *out, int32_t max_results, int32_t *n_out) {
  if (!g || !n_out || (max_results > 0 && !out)) return crn::fail(CRN_ERR_ARG, "crn_ingest_poll: null argument");
  *n_out = 0;
  {  // a hand-off that found both buffers busy is retried here as well as by the next push
    Batch &b = g->batch[g->fill];
    if (is_free(b) && launch_due(g, b)) {
      const int rc = launch(g);
      if (rc != CRN_OK && rc != CRN_ERR_BUSY) return rc;
    }
  }
  int n = 0;
  if (!g->poll_hint.load(std::memory_order_acquire)) return CRN_OK;   // nothing has come back since the last poll: no lock
  {
    std::lock_guard<std::mutex> lk(g->mu);
    if (g->err_code != CRN_OK) {
      const int code = g->err_code;
      g->err_code = CRN_OK;
      g->poll_hint.store(!g->ready.empty(), std::memory_order_release);
      return crn::fail(code, g->err_msg);
    }
    while (n < max_results && !g->ready.empty()) {
      out[n++] = g->ready.front();
      g->ready.pop_front();
    }
    g->poll_hint.store(!g->ready.empty(), std::memory_order_release);
  }
  g->polled += n;
  *n_out = n;
  return CRN_OK;
}

int crn_ingest_drain(crn_ingest *g) {
  if (!g) return crn::fail(CRN_ERR_ARG, "null ingest ring");
  if (int rc = crn_ingest_flush(g)) return rc;
  wait_free(g, g->batch[0]);
  wait_free(g, g->batch[1]);
  return sticky_error(g);
}

int crn_ingest_dropped(crn_ingest *g, int64_t *n_packets) {
  if (!g || !n_packets) return crn::fail(CRN_ERR_ARG, "null argument");
  *n_packets = g->dropped;
  return CRN_OK;
}

int crn_ingest_get_stats(crn_ingest *g, crn_ingest_stats *out) {
  if (!g || !out) return crn::fail(CRN_ERR_ARG, "null argument");
  out->packets = g->packets;
  out->dropped = g->dropped;
  out->epochs_polled = g->polled;
  std::lock_guard<std::mutex> lk(g->mu);
  out->batches = g->n_batches;
  out->batches_failed = g->n_failed;
  out->epochs_launched = g->n_epochs_launched;
  out->epochs_ready = g->n_epochs_ready;
  out->latency_us_sum = g->lat_us_sum;
  out->latency_us_max = g->lat_us_max;
  return CRN_OK;
}

int crn_ingest_destroy(crn_ingest *g) {
  if (!g) return CRN_OK;
  if (g->launcher.joinable()) {
    {
      std::lock_guard<std::mutex> lk(g->mu);
      g->stop = true;
    }
    g->cv_work.notify_all();
    g->launcher.join();  // launches what is queued, waits for what is in flight
  }
  if (g->trace && g->tr_n > 0)
    std::fprintf(stderr, "crn_ingest trace: %lld batches; hand-off -> launcher has it %.1f us, -> launch enqueued %.1f us, -> results seen %.1f us (means)\n",
                 (long long)g->tr_n, g->tr_wake_us / g->tr_n, g->tr_enqueue_us / g->tr_n, g->tr_device_us / g->tr_n);
  (void)hipSetDevice(g->cfg.device);
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
  if (g->attached) (void)crn_sense_ring_count(g->h, -1);
  delete g;
  return CRN_OK;
}

}  // extern "C"
