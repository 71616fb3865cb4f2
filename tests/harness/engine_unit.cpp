// engine_unit.cpp — TEST INFRASTRUCTURE.  The engine's HOST logic without a GPU: CE_Predictive_Node_GPU.cpp + the real ingest
// ring (csrc/crn_ingest.cpp) + the real configuration helpers (csrc/crn_cfg.cpp) against the host-only stand-ins for the HIP
// runtime and the sensing launch (fake_hip, fake_sense.h), under ThreadSanitizer.  What is checked is the control flow of
// execute() — the reference's (CE_Predictive_Node.cpp:54-292) — not arithmetic: the stand-in "decides" the rounded first sample
// of an epoch, so the test chooses every decision and watches what the engine does with it:
//   first call configuration (.cpp:66-69), the wall-clock gate (.cpp:127-141; >= 100 ms between re-arms), set_ce_sensing(0) on
//   the 10th packet (.cpp:159), set_tx_freq mapping 1 -> 835 MHz, 2 -> 833 MHz, 3 -> 835 MHz, 0 -> no call (.cpp:245-261),
//   decisions reported by a later execute() in the enqueue-only mode, packets longer than the FFT truncated, a packet-length
//   change between epochs, a refused packet (both buffers "on the GPU") skipped without blocking, and the synchronous mode.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
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
extern "C" {
int crn_sense_create(const crn_cfg *cfg, crn_handle **out) {
  *out = new crn_handle();
  (*out)->cfg = *cfg;
  return CRN_OK;
}
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
int crn_sense_run_host(crn_handle *h, const float *iq, int64_t n_epochs, int32_t L, int64_t, const crn_out *o) {
  return crn_sense_run_device(h, iq, n_epochs, L, 0, o, NULL);
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

static void exec(ECRd &ecr, ECRd::CE_Event ev) {
  ecr.CE_metrics.CE_event = ev;
  ecr.CE->execute();
}

static int count(const ECRd &ecr, const char *name, double arg, size_t from = 0) {
  int n = 0;
  for (size_t i = from; i < ecr.calls.size(); i++) n += ecr.calls[i].name == name && ecr.calls[i].arg == arg;
  return n;
}

// Feed one epoch whose stand-in decision is `d`; returns the index in ecr.calls where the epoch began.
static size_t feed_epoch(ECRd &ecr, CE_Predictive_Node_GPU *e, std::vector<std::complex<float> > &buf, int d, bool wait_sensing = true) {
  while (wait_sensing && !ecr.ce_sensing_flag) exec(ecr, ECRd::TIMEOUT);
  const size_t mark = ecr.calls.size();
  const long closed = e->epochs_closed;
  for (int p = 0; p < 10; p++) {
    for (size_t i = 0; i < buf.size(); i++) buf[i] = std::complex<float>((float)d, 0.25f);
    do {
      e->packets_dropped = 0;
      exec(ecr, ECRd::USRP_RX_SAMPS);
    } while (e->packets_dropped);   // a refused packet is offered again (nothing else to do offline)
    if (p < 9) REQUIRE(ecr.ce_sensing_flag == 1);
  }
  REQUIRE(ecr.ce_sensing_flag == 0);                       // .cpp:159: the 10th packet switches sensing off
  for (int spin = 0; e->epochs_closed == closed && spin < 2000000; spin++) exec(ecr, ECRd::TIMEOUT);
  REQUIRE(e->epochs_closed == closed + 1 && e->decision == d);
  return mark;
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
    const long long t_end = fake_hip_now_ns() + 40000000;   // 40 ms of packets as fast as execute() returns
    while (fake_hip_now_ns() < t_end) {
      e->packets_dropped = 0;
      const long long t0 = fake_hip_now_ns();
      exec(ecr, ecr.ce_sensing_flag ? ECRd::USRP_RX_SAMPS : ECRd::TIMEOUT);
      const long dt = (long)(fake_hip_now_ns() - t0);
      if (dt > worst_ns) worst_ns = dt;
      refused += e->packets_dropped;
    }
    REQUIRE(refused > 0);                 // the third epoch could not be staged while two were "on the GPU"
    REQUIRE(worst_ns < 2000000);          // and no call sat out a 3 ms batch
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
  printf("engine_unit: ok\n");
  return 0;
}
