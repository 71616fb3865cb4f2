// engine_harness.cpp — TEST INFRASTRUCTURE.  Offline driver: plays the rx worker + CE worker of the ECR
// (reference: src/extensible_cognitive_radio.cpp:1310-1324 hand-off, :1792-1803 dispatch) against
// CE_Predictive_Node_GPU, feeding packets from a binary file of interleaved fp32 IQ.
//
//   engine_harness [--realtime] <iq.bin> <samples_per_packet> [ce args...]
// prints one line per epoch:  epoch <e> decision <d> tx <freq> feat <4 floats> out <3 doubles>
// then the recorded setter-call sequence, the times of the set_ce_sensing(1) calls, and the
// distribution of the time spent inside execute() (the CE worker holds CE_mutex for that long).
//
// Default pacing: a packet is delivered as soon as the engine has asked for samples; at an epoch end
// the CE worker's TIMEOUT events keep coming until the decision has been reported.
// --realtime: the rx worker only forwards packets while ce_sensing_flag is set (:1310) and the CE
// worker spins on TIMEOUT events otherwise (ce_timeout_ms = 0, scenarios/predictive_model.cfg:61),
// so an engine with its wall-clock gate on senses once per 100 ms as in the field.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

#include <algorithm>
#include <vector>

#include "CE_Predictive_Node_GPU.hpp"

// The plugin base class has three empty members (reference: src/cognitive_engine.cpp:4-6).  The
// harness provides them unless it is linked with the reference's own object code
// (engine_harness_refbase, -DCRN_USE_REFERENCE_BASE).
#ifndef CRN_USE_REFERENCE_BASE
CognitiveEngine::CognitiveEngine() : ECR(NULL) {}
CognitiveEngine::~CognitiveEngine() {}
void CognitiveEngine::execute() { /* engines override this */ }
#endif

static std::vector<float> g_exec_us;
static std::vector<float> g_launch_us;  // the calls that took the K-th packet of an epoch (they enqueue the GPU work)
static float g_control_max_us = 0.f;    // control: the longest gap between two clock reads with nothing between them (the OS alone)

static void timed_execute(ExtensibleCognitiveRadio &ecr) {
  struct timespec a, b, c;
  clock_gettime(CLOCK_MONOTONIC, &a);
  ecr.CE->execute();  // :1802
  clock_gettime(CLOCK_MONOTONIC, &b);
  clock_gettime(CLOCK_MONOTONIC, &c);
  g_exec_us.push_back((float)((b.tv_sec - a.tv_sec) * 1e6 + (b.tv_nsec - a.tv_nsec) * 1e-3));
  const float ctl = (float)((c.tv_sec - b.tv_sec) * 1e6 + (c.tv_nsec - b.tv_nsec) * 1e-3);
  if (ctl > g_control_max_us) g_control_max_us = ctl;
}

static void print_new_epochs(ExtensibleCognitiveRadio &ecr, CE_Predictive_Node_GPU *engine, long *seen) {
  if (engine->epochs_closed == *seen) return;
  // epochs that closed before the newest one inside the same execute() (-b > 1): their decisions only
  for (long e = *seen; e + 1 < engine->epochs_closed; e++)
    if (engine->epochs_closed - e <= 64) printf("epoch %ld decision %d\n", e, engine->recent_decisions[e % 64]);
  *seen = engine->epochs_closed;
  double tx = 0.0;
  for (size_t i = ecr.calls.size(); i-- > 0;) {
    if (ecr.calls[i].name == "set_ce_sensing" && ecr.calls[i].arg == 0.0) break;
    if (ecr.calls[i].name == "set_tx_freq") { tx = ecr.calls[i].arg; break; }
  }
  printf("epoch %ld decision %d tx %.1f feat %.9g %.9g %.9g %.9g out %.17g %.17g %.17g\n", *seen - 1,
         engine->decision, tx, engine->features[0], engine->features[1], engine->features[2],
         engine->features[3], engine->outputs[0], engine->outputs[1], engine->outputs[2]);
}

int main(int argc, char **argv) {
  bool realtime = false;
  int a0 = 1;
  if (argc > 1 && strcmp(argv[1], "--realtime") == 0) {
    realtime = true;
    a0 = 2;
  }
  if (argc < a0 + 2) {
    fprintf(stderr, "usage: %s [--realtime] iq.bin samples_per_packet [ce args]\n", argv[0]);
    return 2;
  }
  const int L = atoi(argv[a0 + 1]);
  FILE *f = fopen(argv[a0], "rb");
  if (!f || L < 1) {
    fprintf(stderr, "cannot open %s\n", argv[a0]);
    return 2;
  }
  ExtensibleCognitiveRadio ecr;
  // set_ce: argv[0] is the program name, ce_args follow (reference: src/crts.cpp:43-81)
  std::vector<char *> ce_argv;
  ce_argv.push_back(argv[0]);
  for (int i = a0 + 2; i < argc; i++) ce_argv.push_back(argv[i]);
  ce_argv.push_back(NULL);
  CE_Predictive_Node_GPU *engine = new CE_Predictive_Node_GPU((int)ce_argv.size() - 1, ce_argv.data(), &ecr);
  ecr.CE = engine;

  // rx worker: one buffer of ce_usrp_rx_buffer_length samples (:1263-1269)
  std::vector<std::complex<float> > buf((size_t)L);
  ecr.ce_usrp_rx_buffer = buf.data();
  ecr.ce_usrp_rx_buffer_length = L;

  // a TIMEOUT event first, as the CE worker delivers before any samples arrive (:1796-1799)
  ecr.CE_metrics.CE_event = ExtensibleCognitiveRadio::TIMEOUT;
  timed_execute(ecr);

  long seen = 0;
  long packets = 0;
  // packets per decision: 10 in the reference (fft_averaging, CE_Predictive_Node.hpp:32); -k and the Welch modes change it
#define PPE (engine->packets_per_epoch())
#define EPOCHS_DONE (engine->epochs_closed + engine->epochs_calibrating)
  if (realtime) {
    bool more = true;
    const double t_end = ecr.now() + 30.0;
    while ((more || EPOCHS_DONE * PPE < packets) && ecr.now() < t_end) {
      if (more && ecr.ce_sensing_flag) {  // :1310: forward a packet only while sensing is on
        more = fread(buf.data(), sizeof(std::complex<float>), (size_t)L, f) == (size_t)L;
        if (!more) continue;
        ecr.CE_metrics.CE_event = ExtensibleCognitiveRadio::USRP_RX_SAMPS;  // :1320
        packets++;
      } else {
        ecr.CE_metrics.CE_event = ExtensibleCognitiveRadio::TIMEOUT;        // :1799
      }
      const bool rx = ecr.CE_metrics.CE_event == ExtensibleCognitiveRadio::USRP_RX_SAMPS;
      timed_execute(ecr);
      packets -= engine->packets_dropped;  // a refused packet is not part of any epoch
      if (rx && !engine->packets_dropped && packets % PPE == 0) g_launch_us.push_back(g_exec_us.back());
      engine->packets_dropped = 0;
      print_new_epochs(ecr, engine, &seen);
    }
  } else {
    while (fread(buf.data(), sizeof(std::complex<float>), (size_t)L, f) == (size_t)L) {
      if (!ecr.ce_sensing_flag) {  // rx worker forwards packets only while sensing is on (:1310)
        ecr.CE_metrics.CE_event = ExtensibleCognitiveRadio::TIMEOUT;
        timed_execute(ecr);
      }
      do {  // a packet the ring refused (both buffers on the GPU) is offered again: offline, nothing is lost
        engine->packets_dropped = 0;
        ecr.CE_metrics.CE_event = ExtensibleCognitiveRadio::USRP_RX_SAMPS;  // :1320
        timed_execute(ecr);
      } while (engine->packets_dropped);
      packets++;
      if (packets % PPE == 0) g_launch_us.push_back(g_exec_us.back());
      // the decision of an epoch lands on a later event: keep the CE worker's TIMEOUT events coming
      // (ce_timeout_ms = 0 in scenarios/predictive_model.cfg:61) at epoch ends
      // (with -b B a launch carries B epochs: their decisions land together after the B-th)
      for (int spin = 0; packets % (PPE * engine->epochs_per_launch()) == 0 && EPOCHS_DONE < packets / PPE && spin < 2000000; spin++) {
        ecr.CE_metrics.CE_event = ExtensibleCognitiveRadio::TIMEOUT;
        timed_execute(ecr);
      }
      print_new_epochs(ecr, engine, &seen);
    }
  }
  fclose(f);
  engine->flush();   // epochs of an incomplete last batch (-b > 1)
  print_new_epochs(ecr, engine, &seen);
  printf("calls");
  for (size_t i = 0; i < ecr.calls.size() && i < 12; i++) printf(" %s(%g)", ecr.calls[i].name.c_str(), ecr.calls[i].arg);
  printf("\n");
  printf("sensing_on_at");
  for (size_t i = 0; i < ecr.calls.size(); i++)
    if (ecr.calls[i].name == "set_ce_sensing" && ecr.calls[i].arg == 1.0) printf(" %.6f", ecr.calls[i].t);
  printf("\n");
  // time inside execute(), first call (one-time radio configuration) excluded
  std::vector<float> v(g_exec_us.begin() + 1, g_exec_us.end());
  std::sort(v.begin(), v.end());
  if (!v.empty()) {
    double sum = 0;
    for (size_t i = 0; i < v.size(); i++) sum += v[i];
    printf("execute_us n %zu mean %.3f median %.3f p99 %.3f p999 %.3f max %.3f\n", v.size(), sum / v.size(),
           v[v.size() / 2], v[(size_t)(v.size() * 0.99)], v[(size_t)(v.size() * 0.999)], v.back());
    printf("control_two_clock_reads_us max %.3f (nothing between the two reads: what the operating system alone does to this thread)\n", g_control_max_us);
  }
  std::sort(g_launch_us.begin(), g_launch_us.end());
  if (!g_launch_us.empty())
    printf("epoch_closing_execute_us n %zu median %.3f p99 %.3f max %.3f\n", g_launch_us.size(),
           g_launch_us[g_launch_us.size() / 2], g_launch_us[(size_t)(g_launch_us.size() * 0.99)], g_launch_us.back());
  engine->release();
  return 0;
}
