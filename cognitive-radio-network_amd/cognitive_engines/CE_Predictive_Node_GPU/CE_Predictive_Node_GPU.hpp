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
// a packet is copied once into the ingest ring's pinned slot, the last packet of an epoch enqueues
// H2D + kernel + D2H on a private HIP stream, and the decision is picked up by a later execute()
// (TIMEOUT events fire continuously: ce_timeout_ms = 0, scenarios/predictive_model.cfg:61).  Every
// allocation happens in the constructor.  `-a 0` selects the synchronous form (decision inside
// the epoch's last execute(), ~45 us) for offline use.
//
// ce_args (scenario file `ce_args = "...";` -> argv, reference: src/crts.cpp:43-81; getopt idiom of
// cognitive_engines/CE_Template/CE_Template.cpp:17-25).  With no arguments the engine is the reference's:
//   -n <fft_len>   512 (default, CE_Predictive_Node.hpp:31) | 1024 | 2048 | 4096; the channel plan keeps its frequency
//                  spans (bin ranges of .cpp:173-191 scaled by n/512)
//   -m <mode>      ref    (default) |X| mean over K frames, square of the band sums, 4-5-3 network, cascade (.cpp:150-261)
//                  energy sum |X|^2 per band; a channel is occupied when it exceeds lambda x its share of the noise-floor band
//                  welch  the same plan and rule on a Hann-windowed, 50 %-overlapped (Welch) estimate; the packets of a
//                         sensing period are taken as one contiguous run of samples
//                  scan   64 equal bands across the whole spectrum on the Welch estimate, per-band absolute threshold
//                         lambda x the noise floor measured over the first -c epochs (median band energy)
//   -k <frames>    frames per decision (default 10, .hpp:32)
//   -w <file>      network weights for -m ref (crn_cfg_load_ann: what crn_ann_train_device fitted for this -n and this
//                  receiver gain; the reference's literals, .cpp:78-120, were fitted at n = 512)
//   -t <lambda>    threshold factor of the energy / welch / scan modes (default 4)
//   -c <epochs>    scan: epochs used to measure the noise floor at start-up (default 8); no decision is acted on meanwhile
//   -b <epochs>    epochs per launch of the enqueue-only path (default 1: every decision as soon as possible)
//   -a 0|1  -d <device>  -g 0|1  -s 0|1  -v 0|1   synchronous form, HIP device, wall-clock gate, counters line, the printf block
#ifndef _CE_PREDICTIVE_NODE_GPU_
#define _CE_PREDICTIVE_NODE_GPU_

#include <sys/time.h>

#include <complex>
#include <vector>

#include "cognitive_engine.hpp"
#include "extensible_cognitive_radio.hpp"
#include "crn_sense.h"

class CE_Predictive_Node_GPU : public CognitiveEngine {
public:
  enum Mode { MODE_REF = 0, MODE_ENERGY = 1, MODE_WELCH = 2, MODE_SCAN = 3 };

private:
  // sensing parameters (reference: CE_Predictive_Node.hpp:30-33,42-43)
  static constexpr float sensing_delay_ms = 1e2;
  static constexpr float Desired_fc = 833e6;
  static constexpr float Desired_BW = 13e6;

  crn_cfg cfg;          // reference constants as data (crn_cfg_reference), or what the ce_args selected
  crn_handle *sensor;   // replaces `fftplan fft` (.hpp:78)
  crn_ingest *ring;     // default path: execute() only enqueues and polls
  int config;           // first-call flag (.hpp:40)
  int fft_counter;      // packets staged in the current epoch (.hpp:46)
  long int sense_time_s, sense_time_us;  // next sensing start (.hpp:37-38)
  bool wall_clock_gate; // -g 0 disables the gettimeofday gate (deterministic offline runs)
  int async_mode;       // -a 0 selects the synchronous form
  int verbose;          // -v 0 silences the reference's printf block
  int stats_on;         // -s 1: time every launch, print one summary line at release()
  int sensing_on;       // what this engine last told set_ce_sensing (the ECR's own flag is private)
  int frame_len;        // samples per staged packet, min(ce_usrp_rx_buffer_length, fft_len)
  int mode;             // Mode
  int epochs_per_batch; // -b
  float lambda;         // -t
  int calib_epochs;     // -c (scan mode)
  int calib_have;       // -a 0: epochs gathered so far; both forms: == calib_epochs once the thresholds are set
  std::vector<float> calib_feat;          // -a 0: [calib_epochs][n_bands], allocated in the constructor (the ring keeps its own)
  unsigned char ch_bands[4][CRN_MAX_BANDS];  // scan: ch_bands[k][b] != 0 when band b overlaps channel k's bins (k = 1..3)

  // -a 0 only: the packets of the running epoch end to end; frames are zero-padded (disjoint) or cut (Welch) by the kernel.
  // Replaces `float _Complex buffer[fft_length]` (.hpp:49): the reference transforms each packet
  // as it arrives; this engine stages the epoch and transforms its K frames in one launch.
  std::vector<std::complex<float> > staging;
  std::vector<std::complex<float> > pad;   // a packet of another length inside an epoch, truncated / zero-padded to frame_len

  int packets_in_epoch(int L) const;       // K, or for overlapped frames ceil(((K - 1) hop + N) / L)
  int channel_decision(const unsigned char *occupancy) const;   // threshold modes: first occupied of CH1, CH2, CH3 (cascade order)
  void close_epoch(const float *feat, const double *out3, int kernel_decision, const unsigned char *occupancy);   // -a 0
  void close_ring_epoch(const crn_epoch_result &r);                                                               // enqueue-only path

public:
  // results of the last closed epoch (the reference only prints them: .cpp:202-261)
  float features[CRN_MAX_BANDS];  // ref / energy / welch: NOISE_FLOOR, CH1, CH2, CH3; scan: the 64 band energies
  double outputs[3];    // Output[1..3] (ref mode)
  int decision;         // 0 = "ALL BUSY", 1..3 = Channel_State[d] OCCUPIED
  int recent_decisions[64];  // decision of epoch e at [e % 64] (with -b > 1 several epochs close inside one execute())
  long epochs_closed;
  long epochs_calibrating;  // scan mode: epochs that went into the noise-floor estimate (not acted on)
  long packets_dropped; // packets the ring refused because both of its buffers were on the GPU
  float noise_floor;    // scan mode: the estimate, once measured

  CE_Predictive_Node_GPU(int argc, char **argv, ExtensibleCognitiveRadio *_ECR);
  ~CE_Predictive_Node_GPU();
  virtual void execute();
  void report(const float *feat, const double *out3, int d);  // print + set_tx_freq (.cpp:202-261)
  int packets_per_epoch() const { return packets_in_epoch(frame_len); }
  int fft_length() const { return cfg.fft_len; }
  int epochs_per_launch() const { return async_mode ? epochs_per_batch : 1; }
  // Launch what is staged, wait for everything in flight and act on those decisions (blocks: for an orderly shutdown or an
  // offline run's last epochs with -b > 1, never from execute()).
  void flush();
  // The ECR never deletes its engine (no `delete CE` in the reference), so GPU resources are
  // released explicitly or at process exit.
  void release();
};

#endif
