// CE_Predictive_Node_GPU — drop-in counterpart of the reference's CE_Predictive_Node
// (cognitive_engines/CE_Predictive_Node/CE_Predictive_Node.{hpp,cpp}) whose per-epoch arithmetic
// runs on an MI355X through libcrnsense's C ABI (crn_sense.h, beside this file).
//
// Same plugin surface: constructor (argc, argv, ECR*), virtual execute(), state in through
// ECR->CE_metrics / ce_usrp_rx_buffer, results out through the ECR setters.  Pure host C++11, so
// CRTS's config_cognitive_engines registrar picks this directory up as it is
// (reference: src/config_cognitive_engines.cpp:43-61,82-134); the only link addition is -lcrnsense.
//
// execute() runs on the CE thread with CE_mutex held while the rx thread waits for that mutex
// (reference: src/extensible_cognitive_radio.cpp:1311,1792-1803), so by default it only ENQUEUES:
// a packet is copied once into the ingest ring's pinned slot, the K-th packet of an epoch enqueues
// H2D + kernel + D2H on a private HIP stream, and the decision is picked up by a later execute()
// (TIMEOUT events fire continuously: ce_timeout_ms = 0, scenarios/predictive_model.cfg:61).  Every
// allocation happens in the constructor.  `-a 0` selects the synchronous form (decision inside
// the K-th execute(), ~45 us) for offline use.
#ifndef _CE_PREDICTIVE_NODE_GPU_
#define _CE_PREDICTIVE_NODE_GPU_

#include <sys/time.h>

#include <complex>
#include <vector>

#include "cognitive_engine.hpp"
#include "extensible_cognitive_radio.hpp"
#include "crn_sense.h"

class CE_Predictive_Node_GPU : public CognitiveEngine {
private:
  // sensing parameters (reference: CE_Predictive_Node.hpp:30-33,42-43)
  static constexpr float sensing_delay_ms = 1e2;
  static constexpr float Desired_fc = 833e6;
  static constexpr float Desired_BW = 13e6;

  crn_cfg cfg;          // reference constants as data (crn_cfg_reference)
  crn_handle *sensor;   // replaces `fftplan fft` (.hpp:78)
  crn_ingest *ring;     // default path: execute() only enqueues and polls
  int config;           // first-call flag (.hpp:40)
  int fft_counter;      // frames staged in the current epoch (.hpp:46)
  long int sense_time_s, sense_time_us;  // next sensing start (.hpp:37-38)
  bool wall_clock_gate; // -g 0 disables the gettimeofday gate (deterministic offline runs)
  int async_mode;       // -a 0 selects the synchronous form
  int verbose;          // -v 0 silences the reference's printf block
  int stats_on;         // -s 1: time every launch, print one summary line at release()
  int sensing_on;       // what this engine last told set_ce_sensing (the ECR's own flag is private)
  int frame_len;        // samples per staged packet, min(ce_usrp_rx_buffer_length, fft_len)

  // -a 0 only: K packets of the running epoch, frame-major, zero-padded per frame by the kernel.
  // Replaces `float _Complex buffer[fft_length]` (.hpp:49): the reference transforms each packet
  // as it arrives; this engine stages the epoch and transforms its K frames in one launch.
  std::vector<std::complex<float> > staging;

public:
  // results of the last closed epoch (the reference only prints them: .cpp:202-261)
  float features[4];    // NOISE_FLOOR, CH1, CH2, CH3
  double outputs[3];    // Output[1..3]
  int decision;         // 0 = "ALL BUSY", 1..3 = Channel_State[d] OCCUPIED
  long epochs_closed;
  long packets_dropped; // packets the ring refused because both of its buffers were on the GPU

  CE_Predictive_Node_GPU(int argc, char **argv, ExtensibleCognitiveRadio *_ECR);
  ~CE_Predictive_Node_GPU();
  virtual void execute();
  void report(const float *feat, const double *out3, int d);  // print + set_tx_freq (.cpp:202-261)
  // The ECR never deletes its engine (no `delete CE` in the reference), so GPU resources are
  // released explicitly or at process exit.
  void release();
};

#endif
