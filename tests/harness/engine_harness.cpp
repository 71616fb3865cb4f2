// engine_harness.cpp — TEST INFRASTRUCTURE.  Offline driver: plays the rx worker + CE worker of the ECR
// (reference: src/extensible_cognitive_radio.cpp:1310-1324 hand-off, :1792-1803 dispatch) against
// CE_Predictive_Node_GPU, feeding packets from a binary file of interleaved fp32 IQ.
//
//   engine_harness <iq.bin> <samples_per_packet> [ce args...]
// prints one line per epoch:  epoch <e> decision <d> tx <freq> feat <4 floats> out <3 doubles>
// and finally the recorded setter-call sequence.
#include <stdio.h>
#include <stdlib.h>
#include <unistd.h>

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

int main(int argc, char **argv) {
  if (argc < 3) {
    fprintf(stderr, "usage: %s iq.bin samples_per_packet [ce args]\n", argv[0]);
    return 2;
  }
  const int L = atoi(argv[2]);
  FILE *f = fopen(argv[1], "rb");
  if (!f || L < 1) {
    fprintf(stderr, "cannot open %s\n", argv[1]);
    return 2;
  }
  ExtensibleCognitiveRadio ecr;
  // set_ce: argv[0] is the program name, ce_args follow (reference: src/crts.cpp:43-81)
  std::vector<char *> ce_argv;
  ce_argv.push_back(argv[0]);
  for (int i = 3; i < argc; i++) ce_argv.push_back(argv[i]);
  ce_argv.push_back(NULL);
  CE_Predictive_Node_GPU *engine = new CE_Predictive_Node_GPU((int)ce_argv.size() - 1, ce_argv.data(), &ecr);
  ecr.CE = engine;

  // rx worker: one buffer of ce_usrp_rx_buffer_length samples (:1263-1269)
  std::vector<std::complex<float> > buf((size_t)L);
  ecr.ce_usrp_rx_buffer = buf.data();
  ecr.ce_usrp_rx_buffer_length = L;

  // a TIMEOUT event first, as the CE worker delivers before any samples arrive (:1796-1799)
  ecr.CE_metrics.CE_event = ExtensibleCognitiveRadio::TIMEOUT;
  ecr.CE->execute();

  long seen = 0;
  long packets = 0;
  while (fread(buf.data(), sizeof(std::complex<float>), (size_t)L, f) == (size_t)L) {
    if (!ecr.ce_sensing_flag) {  // rx worker forwards packets only while sensing is on (:1310)
      ecr.CE_metrics.CE_event = ExtensibleCognitiveRadio::TIMEOUT;
      ecr.CE->execute();
    }
    ecr.CE_metrics.CE_event = ExtensibleCognitiveRadio::USRP_RX_SAMPS;  // :1320
    ecr.CE->execute();                                                  // :1802
    packets++;
    // in the asynchronous mode (-a 1) a decision lands on a later event: keep the CE worker's
    // TIMEOUT events coming (ce_timeout_ms = 0 in scenarios/predictive_model.cfg:61) at epoch ends
    for (int spin = 0; packets % 10 == 0 && engine->epochs_closed < packets / 10 && spin < 20000; spin++) {
      usleep(50);
      ecr.CE_metrics.CE_event = ExtensibleCognitiveRadio::TIMEOUT;
      ecr.CE->execute();
    }
    if (engine->epochs_closed != seen) {
      seen = engine->epochs_closed;
      double tx = 0.0;
      for (size_t i = ecr.calls.size(); i-- > 0;) {
        if (ecr.calls[i].name == "set_ce_sensing" && ecr.calls[i].arg == 0.0) break;
        if (ecr.calls[i].name == "set_tx_freq") { tx = ecr.calls[i].arg; break; }
      }
      printf("epoch %ld decision %d tx %.1f feat %.9g %.9g %.9g %.9g out %.17g %.17g %.17g\n", seen - 1,
             engine->decision, tx, engine->features[0], engine->features[1], engine->features[2],
             engine->features[3], engine->outputs[0], engine->outputs[1], engine->outputs[2]);
    }
  }
  fclose(f);
  printf("calls");
  for (size_t i = 0; i < ecr.calls.size() && i < 12; i++) printf(" %s(%g)", ecr.calls[i].name.c_str(), ecr.calls[i].arg);
  printf("\n");
  engine->release();
  return 0;
}
