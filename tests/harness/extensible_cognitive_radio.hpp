// extensible_cognitive_radio.hpp — TEST DOUBLE: the slice of the ECR that a sensing engine touches.
// (Lives under tests/: in a CRTS tree the engine compiles against the real header.)
//
// The reference's ExtensibleCognitiveRadio (include/extensible_cognitive_radio.hpp, 1000+ lines)
// owns two UHD USRP handles, the liquid-dsp OFDM PHY, a TUN device and three pthreads; none of
// that is on the sensing path and none of it can exist on a GPU box without a radio.  This header
// declares, with the reference's names, types and enumerator values, exactly the members
// CE_Predictive_Node reads and the setters it calls, so the same engine source compiles against
// this harness (offline, tests, benchmarks) or against the real ECR (CRTS build):
//
//   CE_Event / metric_s::CE_event / CE_metrics   include/extensible_cognitive_radio.hpp:65-91,161-169,538
//   ce_usrp_rx_buffer, ce_usrp_rx_buffer_length   :547,550 (filled by the rx worker,
//                                                 src/extensible_cognitive_radio.cpp:1310-1324)
//   set_ce_sensing                                :543, src/...cpp:389-391
//   set_tx_freq / stop_tx                         src/...cpp:528-535, 514-518
//   set_rx_freq / set_rx_rate                     src/...cpp:961-968, 1021-1027
//
// The harness records every setter call so tests can compare the call sequence with the
// reference's (CE_Predictive_Node.cpp:66-69,133-134,159,247,252,257).
#ifndef _ECR_HARNESS_HPP_
#define _ECR_HARNESS_HPP_

#include <time.h>

#include <complex>
#include <string>
#include <vector>

class CognitiveEngine;

class ExtensibleCognitiveRadio {
public:
  enum CE_Event {
    TIMEOUT = 0,
    PHY_FRAME_RECEIVED,
    TX_COMPLETE,
    UHD_OVERFLOW,
    UHD_UNDERRUN,
    USRP_RX_SAMPS  // = 5: "custom spectrum sensing" samples are in ce_usrp_rx_buffer
  };

  struct metric_s {
    ExtensibleCognitiveRadio::CE_Event CE_event;
  };

  struct Call {
    std::string name;
    double arg;
    double t;  // seconds since the double was constructed
  };

  ExtensibleCognitiveRadio() : ce_usrp_rx_buffer(nullptr), ce_usrp_rx_buffer_length(0), CE(nullptr), ce_sensing_flag(0) {
    CE_metrics.CE_event = TIMEOUT;
    clock_gettime(CLOCK_MONOTONIC, &t0);
  }
  double now() const {
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (double)(t.tv_sec - t0.tv_sec) + 1e-9 * (double)(t.tv_nsec - t0.tv_nsec);
  }

  struct metric_s CE_metrics;
  std::complex<float> *ce_usrp_rx_buffer;
  int ce_usrp_rx_buffer_length;

  void set_ce_sensing(int ce_sensing) {
    ce_sensing_flag = ce_sensing;
    calls.push_back(Call{"set_ce_sensing", (double)ce_sensing, now()});
  }
  void set_tx_freq(double f) { calls.push_back(Call{"set_tx_freq", f, now()}); }
  void stop_tx() { calls.push_back(Call{"stop_tx", 0.0, now()}); }
  void set_rx_freq(double f) { calls.push_back(Call{"set_rx_freq", f, now()}); }
  void set_rx_rate(double r) { calls.push_back(Call{"set_rx_rate", r, now()}); }

  // harness side
  CognitiveEngine *CE;
  int ce_sensing_flag;
  std::vector<Call> calls;
  struct timespec t0;
};

#endif
