// engine_unit.cpp — TEST INFRASTRUCTURE.  The engine's HOST logic without a GPU: CE_Predictive_Node_GPU.cpp + the real ingest
// ring (csrc/crn_ingest.cpp) + the real configuration helpers (csrc/crn_cfg.cpp) against the host-only stand-ins for the HIP
// runtime and the sensing launch (fake_hip, fake_sense.h), under ThreadSanitizer.  What is checked is the control flow of
// execute() — the reference's (CE_Predictive_Node.cpp:54-292) — not arithmetic: the stand-in "decides" the rounded first sample
// of an epoch, so the test chooses every decision and watches what the engine does with it:
//   first call configuration (.cpp:66-69), the wall-clock gate (.cpp:127-141; >= 100 ms between re-arms), set_ce_sensing(0) on
//   the 10th packet (.cpp:159), set_tx_freq mapping 1 -> 835 MHz, 2 -> 833 MHz, 3 -> 835 MHz, 0 -> no call (.cpp:245-261),
//   decisions reported by a later execute() in the enqueue-only mode, packets longer than the FFT truncated, a packet-length
//   change between epochs and INSIDE an epoch, a refused packet (both buffers "on the GPU") skipped without blocking — shown by
//   counting: no HIP call of any kind is made from inside execute() —, the synchronous mode, a launch failure on the ring's
//   launcher thread ending the run like the synchronous form's, and the ce_args: -n / -m / -k / -t / -b / -w / -c (FFT size,
//   energy / Welch / scan plans, frames per decision, thresholds, batch size, weights file, noise-floor calibration).
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/wait.h>
#include <unistd.h>

#include <vector>

#include "CE_Predictive_Node_GPU.hpp"
#include "fake_sense.h"

// the engines are deleted through their own (most derived) type below; the base class has a non-virtual destructor, like the reference's
#pragma GCC diagnostic ignored "-Wdelete-non-virtual-dtor"


CognitiveEngine::CognitiveEngine() : ECR(NULL) {}
CognitiveEngine::~CognitiveEngine() {}
void CognitiveEngine::execute() {}

// the rest of libcrnsense the engine links against, restated (the ring and crn_cfg_* are the real ones)
static crn_cfg g_created_cfg;          // what the engine's constructor asked for
extern "C" {
int crn_sense_create(const crn_cfg *cfg, crn_handle **out) {
  *out = new crn_handle();
  (*out)->cfg = *cfg;
  g_created_cfg = *cfg;
  return CRN_OK;
}
static int g_fake_abi_version = CRN_ABI_VERSION;
int crn_abi_version(void) { return g_fake_abi_version; }
int crn_sense_destroy(crn_handle *h) { delete h; return CRN_OK; }
int crn_sense_reserve_host(crn_handle *, int64_t, int32_t) { return CRN_OK; }
static int g_timing_requests = 0;
int crn_sense_set_timing(crn_handle *, int32_t on) { g_timing_requests += on; return CRN_OK; }
int crn_sense_get_stats(crn_handle *, crn_sense_stats *out) {
  memset(out, 0, sizeof(*out));
  out->launches = out->timed_launches = 7;
  out->kernel_ms = 0.07;
  return CRN_OK;
}
int crn_sense_run_host(crn_handle *h, const float *iq, int64_t n_epochs, int32_t L, int64_t stride, const crn_out *o) {
  return crn_sense_run_device(h, iq, n_epochs, L, stride, o, NULL);
}
}

#define REQUIRE(c)                                                              \
  do {                                                                          \
    if (!(c)) {                                                                 \
      fprintf(stderr, "engine_unit: line %d: %s FAILED\n", __LINE__, #c);      \
      exit(1);                                                                  \
    }                                                                           \
  } while (0)

typedef ExtensibleCognitiveRadio ECRd;

// execute() runs with CE_mutex held and must only enqueue: in the enqueue-only mode every call is watched by the HIP stand-in
// (g_watch_execute), which counts any HIP call made from this thread meanwhile
static bool g_watch_execute = false;
static void exec(ECRd &ecr, ECRd::CE_Event ev) {
  ecr.CE_metrics.CE_event = ev;
  fake_hip_watch_thread = g_watch_execute;
  ecr.CE->execute();
  fake_hip_watch_thread = false;
}

static int count(const ECRd &ecr, const char *name, double arg, size_t from = 0) {
  int n = 0;
  for (size_t i = from; i < ecr.calls.size(); i++) n += ecr.calls[i].name == name && ecr.calls[i].arg == arg;
  return n;
}

// Feed one epoch whose stand-in decision is `d`; returns the index in ecr.calls where the epoch began.
static size_t feed_epoch(ECRd &ecr, CE_Predictive_Node_GPU *e, std::vector<std::complex<float> > &buf, int d, bool wait_sensing = true,
                         int P = 10, bool calibrating = false) {
  while (wait_sensing && !ecr.ce_sensing_flag) exec(ecr, ECRd::TIMEOUT);
  const size_t mark = ecr.calls.size();
  const long closed = e->epochs_closed + e->epochs_calibrating;
  for (int p = 0; p < P; p++) {
    for (size_t i = 0; i < buf.size(); i++) buf[i] = std::complex<float>((float)d, 0.25f);
    do {
      e->packets_dropped = 0;
      exec(ecr, ECRd::USRP_RX_SAMPS);
    } while (e->packets_dropped);   // a refused packet is offered again (nothing else to do offline)
    if (p < P - 1) REQUIRE(ecr.ce_sensing_flag == 1);
  }
  REQUIRE(ecr.ce_sensing_flag == 0);                       // .cpp:159: the epoch's last packet switches sensing off
  for (int spin = 0; e->epochs_closed + e->epochs_calibrating == closed && spin < 2000000; spin++) exec(ecr, ECRd::TIMEOUT);
  REQUIRE(e->epochs_closed + e->epochs_calibrating == closed + 1);
  if (!calibrating) REQUIRE(e->decision == d);
  return mark;
}

static int tx_calls(const ECRd &ecr, size_t from, double *last) {
  int n = 0;
  for (size_t i = from; i < ecr.calls.size(); i++)
    if (ecr.calls[i].name == "set_tx_freq") { n++; *last = ecr.calls[i].arg; }
  return n;
}

// Run `body` in a child process and return its exit status (the engine ends the run with exit(EXIT_FAILURE), like the reference:
// src/crts.cpp:111-115).  No thread of this process is alive when it forks (every earlier engine has been released).
template <class F>
static int exit_status_of(F body) {
  fflush(stdout);
  const pid_t pid = fork();
  if (pid == 0) {
    if (!freopen("/dev/null", "w", stderr)) _exit(99);
    body();
    _exit(0);
  }
  int st = 0;
  REQUIRE(waitpid(pid, &st, 0) == pid);
  return WIFEXITED(st) ? WEXITSTATUS(st) : 1000 + WTERMSIG(st);
}

static CE_Predictive_Node_GPU *make_engine(ECRd &ecr, std::vector<const char *> args) {
  static std::vector<std::vector<char> > keep;   // argv strings must outlive getopt
  std::vector<char *> argv;
  args.insert(args.begin(), "engine_unit");
  for (const char *a : args) {
    keep.push_back(std::vector<char>(a, a + strlen(a) + 1));
    argv.push_back(keep.back().data());
  }
  argv.push_back(NULL);
  CE_Predictive_Node_GPU *e = new CE_Predictive_Node_GPU((int)argv.size() - 1, argv.data(), &ecr);
  ecr.CE = e;
  return e;
}

int main() {
  g_fake_decision_from_data = 1;
  const double tx_for[4] = {0.0, 835e6, 833e6, 835e6};
  for (int sync = 0; sync < 2; sync++) {
    // ---- gate off: continuous sensing --------------------------------------------------------------------------------
    ECRd ecr;
    char a0[] = "engine_unit", a1[] = "-g", a2[] = "0", a3[] = "-v", a4[] = "0", a5[] = "-a", a6[] = "0";
    char *argv_async[] = {a0, a1, a2, a3, a4, NULL}, *argv_sync[] = {a0, a1, a2, a3, a4, a5, a6, NULL};
    CE_Predictive_Node_GPU *e = new CE_Predictive_Node_GPU(sync ? 7 : 5, sync ? argv_sync : argv_async, &ecr);
    ecr.CE = e;
    REQUIRE(g_created_cfg.fft_len == 512 && g_created_cfg.frames_per_epoch == 10 && g_created_cfg.mode == CRN_MODE_REF_MAG &&
            g_created_cfg.decide == CRN_DECIDE_ANN && g_created_cfg.n_segs == 5);   // no ce_args: the reference's constants
    g_watch_execute = !sync;                       // enqueue-only mode: not one HIP call from inside execute()
    const long long hip_calls_before = fake_hip_calls_on_watched_threads.load();
    std::vector<std::complex<float> > buf(364);
    ecr.ce_usrp_rx_buffer = buf.data();
    ecr.ce_usrp_rx_buffer_length = 364;
    exec(ecr, ECRd::TIMEOUT);
    // CE_Predictive_Node.cpp:66-69 then the first sensing request
    REQUIRE(ecr.calls.size() == 5 && ecr.calls[0].name == "stop_tx" && ecr.calls[1].name == "set_rx_freq" && ecr.calls[1].arg == 833e6 &&
            ecr.calls[2].name == "set_rx_rate" && ecr.calls[2].arg == 13e6 && ecr.calls[3].name == "stop_tx" &&
            ecr.calls[4].name == "set_ce_sensing" && ecr.calls[4].arg == 1.0);
    for (int rep = 0; rep < 3; rep++)
      for (int d = 0; d <= 3; d++) {
        const size_t mark = feed_epoch(ecr, e, buf, d);
        int tx = 0;
        for (size_t i = mark; i < ecr.calls.size(); i++)
          if (ecr.calls[i].name == "set_tx_freq") { tx++; REQUIRE(ecr.calls[i].arg == tx_for[d]); }
        REQUIRE(tx == (d == 0 ? 0 : 1));                   // "ALL BUSY" tunes nothing (.cpp:260-261)
        REQUIRE(e->features[1] == (float)d && e->outputs[2] == 2.0);
      }
    // a packet longer than the FFT is truncated to 512 samples (the reference overruns its buffer, .cpp:149)
    std::vector<std::complex<float> > big(600);
    ecr.ce_usrp_rx_buffer = big.data();
    ecr.ce_usrp_rx_buffer_length = 600;
    feed_epoch(ecr, e, big, 2);
    REQUIRE(e->features[0] == 10.0f * 512 * (2.0f + 0.25f));   // the checksum of 10 x 512 samples, not 600
    // and a shorter packet size from the next epoch on
    std::vector<std::complex<float> > small(100);
    ecr.ce_usrp_rx_buffer = small.data();
    ecr.ce_usrp_rx_buffer_length = 100;
    feed_epoch(ecr, e, small, 3);
    REQUIRE(e->features[0] == 10.0f * 100 * (3.0f + 0.25f));
    // a different length INSIDE an epoch neither stalls the epoch nor changes its length: 3 packets of 364 set it, then two of
    // 512 (truncated to 364) and five of 100 (zero-padded) complete it — ten packets, one decision
    {
      while (!ecr.ce_sensing_flag) exec(ecr, ECRd::TIMEOUT);
      const long closed = e->epochs_closed;
      std::vector<std::complex<float> > p364(364, std::complex<float>(1.f, 0.25f)), p512(512, std::complex<float>(1.f, 0.25f)),
          p100(100, std::complex<float>(1.f, 0.25f));
      const struct { std::vector<std::complex<float> > *v; int n; } plan[3] = {{&p364, 3}, {&p512, 2}, {&p100, 5}};
      for (int q = 0; q < 3; q++)
        for (int i = 0; i < plan[q].n; i++) {
          ecr.ce_usrp_rx_buffer = plan[q].v->data();
          ecr.ce_usrp_rx_buffer_length = (int)plan[q].v->size();
          do {
            e->packets_dropped = 0;
            exec(ecr, ECRd::USRP_RX_SAMPS);
          } while (e->packets_dropped);
        }
      REQUIRE(ecr.ce_sensing_flag == 0);
      for (int spin = 0; e->epochs_closed == closed && spin < 2000000; spin++) exec(ecr, ECRd::TIMEOUT);
      REQUIRE(e->epochs_closed == closed + 1 && e->decision == 1);
      REQUIRE(e->features[0] == (5 * 364 + 5 * 100) * 1.25f);
    }
    REQUIRE(fake_hip_calls_on_watched_threads.load() == hip_calls_before);   // (sync mode is not watched: it launches from execute())
    g_watch_execute = false;
    e->release();
    delete e;   // (the ECR never deletes its engine; here the destructor path — release() again — runs under the sanitizers)
  }
  // ---- a slow "GPU": packets are refused while both buffers are busy; execute() never waits -------------------------------
  {
    g_fake_gpu_latency_ns = 3000000;   // 3 ms per epoch
    ECRd ecr;
    char a0[] = "engine_unit", a1[] = "-g", a2[] = "0", a3[] = "-v", a4[] = "0";
    char *argv[] = {a0, a1, a2, a3, a4, NULL};
    CE_Predictive_Node_GPU *e = new CE_Predictive_Node_GPU(5, argv, &ecr);
    ecr.CE = e;
    std::vector<std::complex<float> > buf(364, std::complex<float>(1.f, 0.25f));
    ecr.ce_usrp_rx_buffer = buf.data();
    ecr.ce_usrp_rx_buffer_length = 364;
    long refused = 0, worst_ns = 0;
    g_watch_execute = true;
    const long long hip_calls_before = fake_hip_calls_on_watched_threads.load();
    const long long t_end = fake_hip_now_ns() + 40000000;   // 40 ms of packets as fast as execute() returns
    while (fake_hip_now_ns() < t_end) {
      e->packets_dropped = 0;
      const long long t0 = fake_hip_now_ns();
      exec(ecr, ecr.ce_sensing_flag ? ECRd::USRP_RX_SAMPS : ECRd::TIMEOUT);
      const long dt = (long)(fake_hip_now_ns() - t0);
      if (dt > worst_ns) worst_ns = dt;
      refused += e->packets_dropped;
    }
    g_watch_execute = false;
    REQUIRE(refused > 0);                 // the third epoch could not be staged while two were "on the GPU"
    // ... and no call waited for it: execute() made no HIP call at all (no event wait, no stream wait, no query) — the
    // refusals above are how it got past the busy buffers.  (A wall-clock bound on the slowest call is only meaningful
    // without a sanitizer slowing every access down on a shared machine: it is printed, and asserted in the plain build only.)
    REQUIRE(fake_hip_calls_on_watched_threads.load() == hip_calls_before && fake_hip_waits_on_watched_threads.load() == 0);
#if !defined(__SANITIZE_THREAD__) && !defined(__SANITIZE_ADDRESS__)
    REQUIRE(worst_ns < 2000000);
#endif
    printf("engine_unit: slow GPU: %ld packets refused, slowest execute() %.1f us, HIP calls inside execute(): 0\n", refused, worst_ns * 1e-3);
    REQUIRE(e->epochs_closed >= 5);
    e->release();
    delete e;   // (the ECR never deletes its engine; here the destructor path — release() again — runs under the sanitizers)
    g_fake_gpu_latency_ns = 0;
  }
  // ---- the wall-clock gate (default arguments): sensing is re-armed no sooner than every 100 ms (.cpp:127-141, .hpp:30) -----------
  {
    ECRd ecr;
    char a0[] = "engine_unit", a1[] = "-v", a2[] = "0", a3[] = "-s", a4[] = "1";   // -s 1: launches timed, summary line at release()
    char *argv[] = {a0, a1, a2, a3, a4, NULL};
    CE_Predictive_Node_GPU *e = new CE_Predictive_Node_GPU(5, argv, &ecr);
    REQUIRE(g_timing_requests == 1);
    ecr.CE = e;
    std::vector<std::complex<float> > buf(364);
    ecr.ce_usrp_rx_buffer = buf.data();
    ecr.ce_usrp_rx_buffer_length = 364;
    exec(ecr, ECRd::TIMEOUT);
    REQUIRE(ecr.ce_sensing_flag == 1);                     // the first request comes at once (sense_time = construction time)
    for (int d = 1; d <= 3; d++) feed_epoch(ecr, e, buf, d);
    std::vector<double> on;
    for (size_t i = 0; i < ecr.calls.size(); i++)
      if (ecr.calls[i].name == "set_ce_sensing" && ecr.calls[i].arg == 1.0) on.push_back(ecr.calls[i].t);
    REQUIRE(on.size() == 3);                               // one request per epoch: none while an epoch was being staged
    for (size_t i = 1; i < on.size(); i++) REQUIRE(on[i] - on[i - 1] >= 0.0999 && on[i] - on[i - 1] < 0.2);
    // every sensing request is preceded by stop_tx (.cpp:133)
    for (size_t i = 0; i < ecr.calls.size(); i++)
      if (ecr.calls[i].name == "set_ce_sensing" && ecr.calls[i].arg == 1.0) REQUIRE(i > 0 && ecr.calls[i - 1].name == "stop_tx");
    REQUIRE(count(ecr, "set_ce_sensing", 0.0) == 3);
    e->release();
    delete e;   // (the ECR never deletes its engine; here the destructor path — release() again — runs under the sanitizers)
  }
  // ---- ce_args: FFT size, energy plan, frames per decision, threshold factor, batch size (src/crts.cpp:43-81) -----------------
  for (int sync = 0; sync < 2; sync++) {
    ECRd ecr;
    std::vector<const char *> args = {"-g", "0", "-v", "0", "-n", "1024", "-m", "energy", "-k", "5", "-t", "6", "-b", "1", "-d", "0"};
    if (sync) { args.push_back("-a"); args.push_back("0"); }
    CE_Predictive_Node_GPU *e = make_engine(ecr, args);
    REQUIRE(g_created_cfg.fft_len == 1024 && g_created_cfg.frames_per_epoch == 5 && g_created_cfg.mode == CRN_MODE_ENERGY &&
            g_created_cfg.decide == CRN_DECIDE_THRESHOLD && g_created_cfg.ref_band == 0 && g_created_cfg.hop == 1024);
    REQUIRE(g_created_cfg.segs[2].lo == 110 && g_created_cfg.segs[2].hi == 170 && g_created_cfg.segs[2].band == 2);   // .cpp:181 x 2
    REQUIRE(g_created_cfg.thresh[2] == 6.0f * 60.0f / 20.0f);                                                       // lambda x bins / NF bins
    REQUIRE(e->fft_length() == 1024);
    std::vector<std::complex<float> > buf(364);
    ecr.ce_usrp_rx_buffer = buf.data();
    ecr.ce_usrp_rx_buffer_length = 364;
    exec(ecr, ECRd::TIMEOUT);
    for (int d = 0; d <= 3; d++) {   // the stand-in marks occupancy[d]; the engine turns it into the reference's cascade + tx map
      const size_t mark = feed_epoch(ecr, e, buf, d, true, 5);
      double f = 0;
      REQUIRE(tx_calls(ecr, mark, &f) == (d == 0 ? 0 : 1) && (d == 0 || f == tx_for[d]));
      REQUIRE(e->features[0] == 5.0f * 364 * ((float)d + 0.25f));
    }
    e->release();
    delete e;
  }
  // ---- -b 3 and a packet length that changes between the epochs of one batch: the batch keeps its length (packets padded), nothing dies ----
  {
    ECRd ecr;
    CE_Predictive_Node_GPU *e = make_engine(ecr, {"-g", "0", "-v", "0", "-b", "3"});
    std::vector<std::complex<float> > p364(364), p300(300);
    ecr.ce_usrp_rx_buffer = p364.data();
    ecr.ce_usrp_rx_buffer_length = 364;
    exec(ecr, ECRd::TIMEOUT);
    auto feed = [&](std::vector<std::complex<float> > &buf, int d) {
      while (!ecr.ce_sensing_flag) exec(ecr, ECRd::TIMEOUT);
      ecr.ce_usrp_rx_buffer = buf.data();
      ecr.ce_usrp_rx_buffer_length = (int)buf.size();
      for (int p = 0; p < 10; p++) {
        for (size_t i = 0; i < buf.size(); i++) buf[i] = std::complex<float>((float)d, 0.25f);
        do {
          e->packets_dropped = 0;
          exec(ecr, ECRd::USRP_RX_SAMPS);
        } while (e->packets_dropped);
      }
    };
    feed(p364, 1);
    feed(p300, 2);     // set_packet_len is refused (an epoch of the batch is staged): padded to 364, the run goes on
    feed(p300, 3);
    for (int spin = 0; e->epochs_closed < 3 && spin < 2000000; spin++) exec(ecr, ECRd::TIMEOUT);
    REQUIRE(e->epochs_closed == 3);
    REQUIRE(e->recent_decisions[0] == 1 && e->recent_decisions[1] == 2 && e->recent_decisions[2] == 3);
    REQUIRE(e->features[0] == 10.0f * 300 * 3.25f);     // the padded epoch's checksum: 300 samples a packet, zeros behind them
    REQUIRE(g_fake_last_L.load() == 364);                // launched at the batch's length
    feed(p300, 1);     // the next batch starts at the new length
    feed(p300, 2);
    feed(p300, 3);
    for (int spin = 0; e->epochs_closed < 6 && spin < 2000000; spin++) exec(ecr, ECRd::TIMEOUT);
    REQUIRE(e->epochs_closed == 6 && g_fake_last_L.load() == 300);
    e->release();
    delete e;
  }
  // ---- -m welch: the packets of a sensing period are one contiguous run; frames of N cut from it with hop N/2 --------------------
  for (int sync = 0; sync < 2; sync++) {
    ECRd ecr;
    std::vector<const char *> args = {"-g", "0", "-v", "0", "-n", "1024", "-m", "welch", "-k", "8"};
    if (sync) { args.push_back("-a"); args.push_back("0"); }
    CE_Predictive_Node_GPU *e = make_engine(ecr, args);
    REQUIRE(g_created_cfg.hop == 512 && g_created_cfg.window == CRN_WINDOW_HANN && g_created_cfg.frames_per_epoch == 8 &&
            g_created_cfg.n_bands == 4 && g_created_cfg.ref_band == 0);
    std::vector<std::complex<float> > buf(364);
    ecr.ce_usrp_rx_buffer = buf.data();
    ecr.ce_usrp_rx_buffer_length = 364;
    exec(ecr, ECRd::TIMEOUT);
    const int P = (7 * 512 + 1024 + 363) / 364;   // 13 packets cover the 4608 samples of eight half-overlapped frames
    for (int d = 1; d <= 3; d++) {
      const size_t mark = feed_epoch(ecr, e, buf, d, true, P);
      double f = 0;
      REQUIRE(tx_calls(ecr, mark, &f) == 1 && f == tx_for[d]);
      REQUIRE(g_fake_last_L.load() == 1024 && g_fake_last_stride.load() == (long long)P * 364);   // whole frames, epoch stride = the run
      REQUIRE(e->features[0] == 4608.0f * ((float)d + 0.25f));
      REQUIRE(e->packets_per_epoch() == P);
    }
    e->release();
    delete e;
  }
  // ---- -m scan: 64 equal bands, thresholds = lambda x the noise floor measured over the first -c epochs ----------------------------
  // Enqueue-only form: the calibration belongs to the ring's launcher thread (crn_ingest_calibrate) — execute() is watched through
  // ALL of it, calibrating epochs included, and makes not one HIP call.  Synchronous form (-a 0, offline): execute() launches and
  // waits anyway; there the calibration is one allocation-free call (crn_sense_calibrate_thresholds after crn_sense_reserve_noise_floor).
  for (int sync = 0; sync < 2; sync++) {
    ECRd ecr;
    std::vector<const char *> args = {"-g", "0", "-v", "0", "-n", "1024", "-m", "scan", "-c", "3", "-t", "5"};
    if (sync) { args.push_back("-a"); args.push_back("0"); }
    const int reserved0 = g_fake_nf_reserved.load();
    CE_Predictive_Node_GPU *e = make_engine(ecr, args);
    REQUIRE(g_created_cfg.n_bands == 64 && g_created_cfg.hop == 512 && g_created_cfg.ref_band == -1 && g_created_cfg.frames_per_epoch == 8);
    REQUIRE(g_fake_nf_reserved.load() > reserved0);           // the calibration's buffers were made by the constructor
    std::vector<std::complex<float> > buf(512);
    ecr.ce_usrp_rx_buffer = buf.data();
    ecr.ce_usrp_rx_buffer_length = 512;
    g_watch_execute = !sync;
    const long long hip_calls_before = fake_hip_calls_on_watched_threads.load();
    exec(ecr, ECRd::TIMEOUT);
    const int P = (7 * 512 + 1024 + 511) / 512;
    const int updates0 = g_fake_threshold_updates.load();
    const size_t mark = ecr.calls.size();
    for (int i = 0; i < 3; i++) feed_epoch(ecr, e, buf, 2, true, P, true);   // calibration epochs: a "decision" of 2 must NOT be acted on
    double f = 0;
    REQUIRE(tx_calls(ecr, mark, &f) == 0 && e->epochs_closed == 0 && e->epochs_calibrating == 3);
    REQUIRE(g_fake_threshold_updates.load() == updates0 + 1);
    REQUIRE(g_fake_thresholds_set[0] == 10.0f && g_fake_thresholds_set[63] == 10.0f);   // lambda x the estimate, every band
    for (int d = 0; d <= 3; d++) {   // afterwards the stand-in's occupied band (the one holding the channel's first bin) maps back to the channel
      const size_t m2 = feed_epoch(ecr, e, buf, d, true, P);
      REQUIRE(tx_calls(ecr, m2, &f) == (d == 0 ? 0 : 1) && (d == 0 || f == tx_for[d]));
      REQUIRE(e->noise_floor == 2.0f);                         // (the enqueue-only form learns it with the first acted-on epoch)
    }
    REQUIRE(g_fake_threshold_updates.load() == updates0 + 1 && e->epochs_calibrating == 3 && e->epochs_closed == 4);
    if (!sync) REQUIRE(fake_hip_calls_on_watched_threads.load() == hip_calls_before && fake_hip_waits_on_watched_threads.load() == 0);
    g_watch_execute = false;
    e->release();
    delete e;
  }
  // ---- scan, a slow "GPU" and -b 2: epochs launched before the measured thresholds were in place are marked and not acted on ------
  {
    g_fake_gpu_latency_ns = 2000000;   // 2 ms per batch: further batches are launched while the last calibration batch is in flight
    ECRd ecr;
    CE_Predictive_Node_GPU *e = make_engine(ecr, {"-g", "0", "-v", "0", "-n", "1024", "-m", "scan", "-c", "2", "-t", "5", "-b", "1"});
    std::vector<std::complex<float> > buf(512, std::complex<float>(3.f, 0.25f));
    ecr.ce_usrp_rx_buffer = buf.data();
    ecr.ce_usrp_rx_buffer_length = 512;
    g_watch_execute = true;
    const long long hip_calls_before = fake_hip_calls_on_watched_threads.load();
    const long long t_end = fake_hip_now_ns() + 60000000;
    while (fake_hip_now_ns() < t_end) exec(ecr, ecr.ce_sensing_flag ? ECRd::USRP_RX_SAMPS : ECRd::TIMEOUT);
    g_watch_execute = false;
    REQUIRE(fake_hip_calls_on_watched_threads.load() == hip_calls_before && fake_hip_waits_on_watched_threads.load() == 0);
    // two epochs fed the estimate; whatever else was launched before the update is counted with them, and every epoch acted on
    // was decided against the measured thresholds
    REQUIRE(e->epochs_calibrating >= 2 && e->epochs_closed >= 1 && e->noise_floor == 3.0f);
    double f = 0;
    REQUIRE(tx_calls(ecr, 0, &f) == (int)e->epochs_closed && f == tx_for[3]);
    printf("engine_unit: scan, slow GPU: %ld epochs marked calibration (2 fed the estimate), %ld acted on, HIP calls inside execute(): 0\n",
           e->epochs_calibrating, e->epochs_closed);
    e->release();
    delete e;
    g_fake_gpu_latency_ns = 0;
  }
  // ---- a library of another ABI version ends the run at construction (struct layouts differ: nothing may be called) -----------------
  {
    g_fake_abi_version = CRN_ABI_VERSION - 1;
    const int st = exit_status_of([] {
      ECRd ecr;
      make_engine(ecr, {"-g", "0", "-v", "0"});
    });
    g_fake_abi_version = CRN_ABI_VERSION;
    REQUIRE(st == EXIT_FAILURE);
  }
  // ---- -w: the trainer's weights reach the engine through a file, at another FFT size ----------------------------------------------
  {
    crn_cfg w;
    REQUIRE(crn_cfg_reference(&w) == CRN_OK);
    for (int i = 0; i < 5; i++)
      for (int j = 0; j < 6; j++) w.ann_w_ih[i][j] = 0.1 * i - 0.37 * j + 1.0 / 3.0;
    for (int j = 0; j < 6; j++)
      for (int k = 0; k < 4; k++) w.ann_w_ho[j][k] = -2.5 * j + 0.7 * k + 1e-9 / 7.0;
    w.ann_threshold = 0.65;
    char path[] = "/tmp/crn_engine_unit_weights_XXXXXX";
    const int fd = mkstemp(path);
    REQUIRE(fd >= 0);
    close(fd);
    REQUIRE(crn_cfg_save_ann(&w, path) == CRN_OK);
    ECRd ecr;
    CE_Predictive_Node_GPU *e = make_engine(ecr, {"-g", "0", "-v", "0", "-n", "2048", "-w", path});
    REQUIRE(g_created_cfg.fft_len == 2048 && g_created_cfg.mode == CRN_MODE_REF_MAG && g_created_cfg.decide == CRN_DECIDE_ANN);
    REQUIRE(memcmp(g_created_cfg.ann_w_ih, w.ann_w_ih, sizeof(w.ann_w_ih)) == 0 && memcmp(g_created_cfg.ann_w_ho, w.ann_w_ho, sizeof(w.ann_w_ho)) == 0);
    REQUIRE(g_created_cfg.ann_threshold == 0.65);
    REQUIRE(g_created_cfg.segs[1].lo == 496 * 4 && g_created_cfg.segs[1].hi == 511 * 4);   // .cpp:177 x 4: bin "511" stays out
    e->release();
    delete e;
    // a damaged file is an error, not a silent fallback to the reference's weights
    FILE *fp = fopen(path, "w");
    REQUIRE(fp != NULL);
    fprintf(fp, "1 2 3 four\n");
    fclose(fp);
    crn_cfg bad;
    crn_cfg_reference(&bad);
    REQUIRE(crn_cfg_load_ann(&bad, path) == CRN_ERR_ARG);
    const std::string p2 = path;
    REQUIRE(exit_status_of([&] {
              ECRd ecr2;
              make_engine(ecr2, {"-v", "0", "-w", p2.c_str()});
            }) == EXIT_FAILURE);
    unlink(path);
  }
  // ---- bad ce_args end the run with a message (the reference's convention: printf + exit, src/crts.cpp:111-115) -----------------------
  REQUIRE(exit_status_of([] { ECRd ecr; make_engine(ecr, {"-n", "1000"}); }) == EXIT_FAILURE);
  REQUIRE(exit_status_of([] { ECRd ecr; make_engine(ecr, {"-m", "fourier"}); }) == EXIT_FAILURE);
  REQUIRE(exit_status_of([] { ECRd ecr; make_engine(ecr, {"-m", "energy", "-w", "/dev/null"}); }) == EXIT_FAILURE);
  // ---- a launch that fails on the ring's launcher thread ends the run, as the same failure does in the synchronous form ----------------
  for (int sync = 0; sync < 2; sync++) {
    const int st = exit_status_of([sync] {
      ECRd ecr;
      std::vector<const char *> args = {"-g", "0", "-v", "0"};
      if (sync) { args.push_back("-a"); args.push_back("0"); }
      CE_Predictive_Node_GPU *e = make_engine(ecr, args);
      std::vector<std::complex<float> > buf(364, std::complex<float>(1.f, 0.25f));
      ecr.ce_usrp_rx_buffer = buf.data();
      ecr.ce_usrp_rx_buffer_length = 364;
      exec(ecr, ECRd::TIMEOUT);
      g_fail_next_launch = 1;
      for (int p = 0; p < 10; p++) exec(ecr, ECRd::USRP_RX_SAMPS);
      for (int spin = 0; spin < 2000000 && e->epochs_closed == 0; spin++) exec(ecr, ECRd::TIMEOUT);
      _exit(e->epochs_closed == 0 ? 42 : 43);   // not reached: the failure is fatal
    });
    REQUIRE(st == EXIT_FAILURE);
  }
  printf("engine_unit: ok\n");
  return 0;
}
