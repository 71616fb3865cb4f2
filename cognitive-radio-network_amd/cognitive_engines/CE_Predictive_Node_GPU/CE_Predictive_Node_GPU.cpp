// CE_Predictive_Node_GPU.cpp — see the header.  Control flow follows the reference's execute()
// (cognitive_engines/CE_Predictive_Node/CE_Predictive_Node.cpp:54-292) statement for statement;
// the arithmetic of lines 148-261 is one launch of libcrnsense per epoch.
#include "CE_Predictive_Node_GPU.hpp"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

static void die_crn(void) {
  // the reference's convention for unrecoverable set-up errors (src/crts.cpp:111-115)
  fprintf(stderr, "CE_Predictive_Node_GPU: %s\n", crn_last_error());
  exit(EXIT_FAILURE);
}

static void die_arg(const char *what, const char *value) {
  fprintf(stderr, "CE_Predictive_Node_GPU: bad ce_args: %s '%s'\n", what, value ? value : "");
  exit(EXIT_FAILURE);
}

// constructor (reference: CE_Predictive_Node.cpp:18-46).  Everything the engine will ever
// allocate — tables, device batch buffers, pinned staging, the HIP stream — is allocated here,
// and the kernels are loaded by one launch over zeros: execute() never allocates.
CE_Predictive_Node_GPU::CE_Predictive_Node_GPU(int argc, char **argv, ExtensibleCognitiveRadio *_ECR) {
  ECR = _ECR;

  fft_counter = 0;
  config = 0;
  sensor = NULL;
  ring = NULL;
  async_mode = 1;
  wall_clock_gate = true;
  verbose = 1;
  frame_len = 0;
  sensing_on = 0;
  decision = 0;
  epochs_closed = 0;
  epochs_calibrating = 0;
  packets_dropped = 0;
  noise_floor = 0.f;
  mode = MODE_REF;
  epochs_per_batch = 1;
  lambda = 4.0f;
  calib_epochs = 8;
  calib_have = 0;
  memset(features, 0, sizeof(features));
  memset(outputs, 0, sizeof(outputs));
  memset(ch_bands, 0, sizeof(ch_bands));
  memset(recent_decisions, 0, sizeof(recent_decisions));

  // the library this process loaded must speak the ABI this file was compiled against (struct layouts: crn_cfg, crn_epoch_result);
  // crn_cfg.abi_version cannot tell — the crn_cfg_* helpers below fill it in with the LIBRARY's version
  if (crn_abi_version() != CRN_ABI_VERSION) {
    fprintf(stderr, "CE_Predictive_Node_GPU: libcrnsense speaks ABI version %d, this engine was built against %d (rebuild one of them)\n",
            crn_abi_version(), CRN_ABI_VERSION);
    exit(EXIT_FAILURE);
  }

  // ce_args from the scenario file arrive as argv (reference: src/crts.cpp:43-81 str2argcargv,
  // which resets optind; CE_Template.cpp:17-25 shows the getopt idiom)
  int o;
  optind = 1;
  stats_on = 0;
  int n_fft = 512, k_frames = 0, device = 0;
  const char *weights = NULL;
  while ((o = getopt(argc, argv, "a:b:c:d:g:k:m:n:s:t:v:w:")) != EOF) {
    switch (o) {
    case 'a': async_mode = atoi(optarg); break;            // 0: decide inside the epoch's last execute()
    case 'b': epochs_per_batch = atoi(optarg); break;      // epochs per launch of the enqueue-only path
    case 'c': calib_epochs = atoi(optarg); break;          // scan: noise-floor epochs
    case 'd': device = atoi(optarg); break;                // HIP device ordinal
    case 'g': wall_clock_gate = atoi(optarg) != 0; break;  // 0: sense continuously
    case 'k': k_frames = atoi(optarg); break;              // frames per decision
    case 'm':
      if (strcmp(optarg, "ref") == 0) mode = MODE_REF;
      else if (strcmp(optarg, "energy") == 0) mode = MODE_ENERGY;
      else if (strcmp(optarg, "welch") == 0) mode = MODE_WELCH;
      else if (strcmp(optarg, "scan") == 0) mode = MODE_SCAN;
      else die_arg("-m expects ref | energy | welch | scan, got", optarg);
      break;
    case 'n': n_fft = atoi(optarg); break;
    case 's': stats_on = atoi(optarg); break;               // 1: time every launch, one summary line at release()
    case 't': lambda = (float)atof(optarg); break;
    case 'v': verbose = atoi(optarg); break;
    case 'w': weights = optarg; break;
    default: die_arg("unknown option", argv[optind > 0 ? optind - 1 : 0]);
    }
  }
  if (n_fft != 512 && n_fft != 1024 && n_fft != 2048 && n_fft != 4096) die_arg("-n expects 512 | 1024 | 2048 | 4096", "");
  if (k_frames < 0 || k_frames > 4096) die_arg("-k out of range", "");
  if (epochs_per_batch < 1 || epochs_per_batch > 4096) die_arg("-b out of range", "");
  if (!(lambda > 0.f)) die_arg("-t must be positive", "");
  if (calib_epochs < 1 || calib_epochs > 4096) die_arg("-c out of range", "");
  if (weights && mode != MODE_REF) die_arg("-w goes with -m ref, not with", "-m energy | welch | scan");

  // the reference's constants (crn_cfg_reference), or the plan the arguments ask for
  int rc = CRN_OK;
  if (mode == MODE_REF) rc = crn_cfg_reference_scaled(&cfg, n_fft);
  else if (mode == MODE_ENERGY) rc = crn_cfg_energy_scaled(&cfg, n_fft, lambda);
  else if (mode == MODE_WELCH) rc = crn_cfg_welch_scaled(&cfg, n_fft, k_frames > 0 ? k_frames : 10, lambda);
  else rc = crn_cfg_welch(&cfg, n_fft, k_frames > 0 ? k_frames : 8, 64);
  if (rc != CRN_OK) die_crn();
  if (k_frames > 0) cfg.frames_per_epoch = k_frames;
  cfg.device = device;
  if (weights && crn_cfg_load_ann(&cfg, weights) != CRN_OK) die_crn();
  if (mode == MODE_REF && n_fft != 512 && !weights && verbose)
    printf("CE_Predictive_Node_GPU: -n %d without -w: the reference's weights were fitted at 512 points\n", n_fft);
  if (mode == MODE_SCAN) {
    // which of the 64 equal bands lie on each channel: the reference's bin ranges (.cpp:173-191) at this FFT size
    crn_cfg plan;
    if (crn_cfg_reference_scaled(&plan, n_fft) != CRN_OK) die_crn();
    const int w = n_fft / cfg.n_bands;
    for (int s = 0; s < plan.n_segs; s++)
      for (int k = plan.segs[s].lo; k < plan.segs[s].hi; k++)
        if (plan.segs[s].band >= 1 && plan.segs[s].band <= 3) ch_bands[plan.segs[s].band][k / w] = 1;
    calib_feat.assign((size_t)calib_epochs * cfg.n_bands, 0.f);
  } else {
    calib_have = calib_epochs = 0;
  }

  struct timeval tv;
  gettimeofday(&tv, NULL);
  sense_time_s = tv.tv_sec;
  sense_time_us = tv.tv_usec;

  // replaces memset of the three buffers + fft_create_plan (.cpp:36-45)
  if (crn_sense_create(&cfg, &sensor) != CRN_OK) die_crn();
  if (crn_sense_reserve_host(sensor, 1, 0) != CRN_OK) die_crn();  // scratch + pinned staging + kernel load
  if (mode == MODE_SCAN && crn_sense_reserve_noise_floor(sensor) != CRN_OK) die_crn();  // the calibration's upload buffers
  if (stats_on && crn_sense_set_timing(sensor, 1) != CRN_OK) die_crn();
  frame_len = cfg.fft_len;
  pad.assign((size_t)cfg.fft_len, std::complex<float>(0.f, 0.f));
  if (async_mode) {
    // one stream; sized for full-length packets, the actual UHD packet length is
    // known only when the rx worker starts (src/extensible_cognitive_radio.cpp:1263-1265)
    if (crn_ingest_create(sensor, 1, cfg.fft_len, epochs_per_batch, &ring) != CRN_OK) die_crn();
    // scan: the first -c epochs that come back feed the noise-floor estimate; the ring's launcher thread reduces them and sets
    // the thresholds (this only posts the request: nothing of it ever runs on the CE thread)
    if (mode == MODE_SCAN && crn_ingest_calibrate(ring, calib_epochs, lambda) != CRN_OK) die_crn();
  } else {
    // the longest run an epoch can need at any packet length: K packets of N (disjoint), span + one packet (overlapped)
    const size_t span = (size_t)(cfg.frames_per_epoch - 1) * cfg.hop + cfg.fft_len;
    staging.assign(cfg.hop == cfg.fft_len ? (size_t)cfg.frames_per_epoch * cfg.fft_len : span + cfg.fft_len, std::complex<float>(0.f, 0.f));
  }
}

// destructor (reference: .cpp:49 — empty, and never run by the ECR)
CE_Predictive_Node_GPU::~CE_Predictive_Node_GPU() { release(); }

void CE_Predictive_Node_GPU::release() {
  if (stats_on && sensor) {  // the counters of the C ABI (the reference has only its per-epoch printf lines)
    crn_sense_stats ss;
    crn_ingest_stats is;
    memset(&ss, 0, sizeof(ss));
    memset(&is, 0, sizeof(is));
    if (ring) {
      crn_ingest_drain(ring);
      crn_ingest_get_stats(ring, &is);
    }
    crn_sense_get_stats(sensor, &ss);
    printf("CE_Predictive_Node_GPU: epochs %lld  launches %lld  samples %lld  packets refused %lld  kernel %.1f us mean (%.1f .. %.1f)",
           (long long)epochs_closed, (long long)ss.launches, (long long)ss.samples, (long long)packets_dropped,
           ss.timed_launches ? 1e3 * ss.kernel_ms / (double)ss.timed_launches : 0.0, 1e3 * ss.kernel_ms_min, 1e3 * ss.kernel_ms_max);
    if (ring)
      printf("  hand-off to decision %.1f us mean, %.1f us max", is.batches ? is.latency_us_sum / (double)is.batches : 0.0,
             is.latency_us_max);
    printf("\n");
  }
  if (ring) crn_ingest_destroy(ring);
  ring = NULL;
  if (sensor) crn_sense_destroy(sensor);
  sensor = NULL;
}

void CE_Predictive_Node_GPU::flush() {
  if (!ring) return;
  if (crn_ingest_drain(ring) != CRN_OK) die_crn();
  crn_epoch_result r;
  int32_t n = 0;
  int rc;
  while ((rc = crn_ingest_poll(ring, &r, 1, &n)) == CRN_OK && n == 1) close_ring_epoch(r);
  if (rc != CRN_OK) die_crn();
}

int CE_Predictive_Node_GPU::packets_in_epoch(int L) const {
  if (cfg.hop == cfg.fft_len || L < 1) return cfg.frames_per_epoch;
  const long span = (long)(cfg.frames_per_epoch - 1) * cfg.hop + cfg.fft_len;
  return (int)((span + L - 1) / L);
}

// Threshold modes: the kernel reports per-band occupancy; the engine's decision keeps the reference's cascade order
// (.cpp:245-258: CH1 first, then CH2, then CH3; at most one channel is ever reported).
int CE_Predictive_Node_GPU::channel_decision(const unsigned char *occ) const {
  for (int k = 1; k <= 3; k++) {
    if (mode != MODE_SCAN) {
      if (occ[k]) return k;      // bands are {NF, CH1, CH2, CH3}
    } else {
      for (int b = 0; b < cfg.n_bands; b++)
        if (ch_bands[k][b] && occ[b]) return k;
    }
  }
  return 0;
}

// One epoch's results from the ring (enqueue-only path).  Scan mode, start-up: epochs launched before the measured thresholds were
// in place come back marked CRN_EPOCH_CALIBRATION — they fed the estimate (crn_ingest_calibrate, on the ring's launcher thread) or
// were decided against the thresholds of before: counted, never acted on.  The first unmarked epoch carries the estimate.
void CE_Predictive_Node_GPU::close_ring_epoch(const crn_epoch_result &r) {
  if (r.flags & CRN_EPOCH_CALIBRATION) {
    epochs_calibrating++;
    return;
  }
  if (calib_have < calib_epochs) {   // scan mode: the calibration has landed
    calib_have = calib_epochs;
    noise_floor = r.noise_floor;
    for (int b = 0; b < cfg.n_bands; b++) cfg.thresh[b] = lambda * noise_floor;   // (this copy is for the report; the device has its own)
    if (verbose) printf("CE_Predictive_Node_GPU: noise floor %.4e per band over %d epochs, threshold %.4e\n", noise_floor, calib_epochs, cfg.thresh[0]);
  }
  report(r.features, r.ann_out, mode == MODE_REF ? r.decision : channel_decision(r.occupancy));
}

// One epoch's results in the synchronous form (-a 0: execute() itself launches and waits, for offline use): scan-mode calibration
// first, then the reference's report + action block.
void CE_Predictive_Node_GPU::close_epoch(const float *feat, const double *out3, int kernel_decision, const unsigned char *occupancy) {
  if (calib_have < calib_epochs) {
    // scan mode, start-up: the thresholds are lambda x the measured noise floor (NF_est = the median band
    // energy), so the first epochs only feed the estimate.  The last of them pays for one upload + reduction + update (no allocation:
    // crn_sense_reserve_noise_floor ran in the constructor).
    memcpy(&calib_feat[(size_t)calib_have * cfg.n_bands], feat, sizeof(float) * (size_t)cfg.n_bands);
    epochs_calibrating++;
    if (++calib_have == calib_epochs) {
      if (crn_sense_calibrate_thresholds(sensor, calib_feat.data(), calib_epochs, lambda, &noise_floor, NULL) != CRN_OK) die_crn();
      for (int b = 0; b < cfg.n_bands; b++) cfg.thresh[b] = lambda * noise_floor;
      if (verbose) printf("CE_Predictive_Node_GPU: noise floor %.4e per band over %d epochs, threshold %.4e\n", noise_floor, calib_epochs, cfg.thresh[0]);
    }
    return;
  }
  const int d = mode == MODE_REF ? kernel_decision : channel_decision(occupancy);
  report(feat, out3, d);
}

// The reference's report + action block (.cpp:202-261) for one closed epoch.
void CE_Predictive_Node_GPU::report(const float *feat, const double *out3, int d) {
  memcpy(features, feat, sizeof(float) * (size_t)cfg.n_bands);
  memcpy(outputs, out3, sizeof(outputs));
  decision = d;
  recent_decisions[epochs_closed % 64] = d;
  epochs_closed++;
  if (verbose) {  // .cpp:202-207, 239-241
    printf("--------------------------------------------------------------\n");
    printf("-            		FEATURES BUFFER 	               -\n");
    printf("--------------------------------------------------------------\n");
    if (mode != MODE_SCAN)
      printf("NOISE FLOOR   %.2e\nCH1           %.2e\nCH2           %.2e\nCH3           %.2e\n ", features[0],
             features[1], features[2], features[3]);
    else
      printf("%d bands, noise floor %.2e, threshold %.2e\n ", cfg.n_bands, noise_floor, cfg.thresh[0]);
    printf("\n \n \n --------------------------------------------------------------\n");
    printf("-            		 REAL TIME PREDICTION                  -\n");
    printf("--------------------------------------------------------------\n");
  }
  if (d >= 1 && d <= 3) {
    if (verbose)
      printf("Channel_State[1]: %s \nChannel_State[2]: %s \nChannel_State[3]: %s \n \n \n",
             d == 1 ? "OCCUPIED" : "FREE", d == 2 ? "OCCUPIED" : "FREE", d == 3 ? "OCCUPIED" : "FREE");
    ECR->set_tx_freq(cfg.tx_freq_for_decision[d]);  // .cpp:247,252,257
  } else if (verbose) {
    printf("ALL BUSY, SENSE AND OBSERVE AGAIN \n");  // .cpp:261
  }
}

void CE_Predictive_Node_GPU::execute() {
  // one-time radio configuration (.cpp:66-69); the weights of .cpp:78-120 live in cfg
  if (config == 0) {
    ECR->stop_tx();
    ECR->set_rx_freq(Desired_fc);
    ECR->set_rx_rate(Desired_BW);
    config = 1;
  }

  // turn sensing on once the delay has passed (.cpp:127-141).  The reference adds the delay to
  // tv_usec without carrying into tv_sec (.cpp:139-140); here the carry is done.
  if (wall_clock_gate) {
    struct timeval tv;
    gettimeofday(&tv, NULL);
    if ((tv.tv_sec > sense_time_s) || ((tv.tv_sec == sense_time_s) && (tv.tv_usec >= sense_time_us))) {
      ECR->stop_tx();
      ECR->set_ce_sensing(1);
      sensing_on = 1;
      long int us = tv.tv_usec + (long int)floorf(sensing_delay_ms * 1e3);
      sense_time_s = tv.tv_sec + us / 1000000;
      sense_time_us = us % 1000000;
    }
  } else if (fft_counter == 0 && !sensing_on) {
    ECR->stop_tx();
    ECR->set_ce_sensing(1);
    sensing_on = 1;
  }

  // a decision launched by an earlier call may have landed (one event query, never a wait).  A launch or device failure on
  // the ring's launcher thread is reported here and nowhere else: it ends the run like the same failure in the
  // synchronous form does (the reference's convention: printf + exit, src/crts.cpp:111-115).
  if (ring) {
    crn_epoch_result r;
    int32_t n = 0;
    int rc;
    while ((rc = crn_ingest_poll(ring, &r, 1, &n)) == CRN_OK && n == 1) close_ring_epoch(r);
    if (rc != CRN_OK) die_crn();
  }

  // handle samples (.cpp:146)
  if (ECR->CE_metrics.CE_event == ExtensibleCognitiveRadio::USRP_RX_SAMPS) {
    const int N = cfg.fft_len;
    // .cpp:149 copies ce_usrp_rx_buffer_length samples unchecked; a packet longer than the FFT
    // would overrun the reference's buffer — here it is truncated to N.
    int L = ECR->ce_usrp_rx_buffer_length;
    if (L > N) L = N;
    if (L < 1) return;
    const std::complex<float> *pkt = ECR->ce_usrp_rx_buffer;
    if (ring) {
      if (L != frame_len && fft_counter == 0) {   // UHD packet size is constant in practice: once, at the first packet
        const int rc = crn_ingest_set_packet_len(ring, L);
        if (rc == CRN_ERR_BUSY) { packets_dropped++; return; }
        if (rc == CRN_OK) frame_len = L;
        else if (rc != CRN_ERR_STATE) die_crn();
        // (CRN_ERR_STATE: with -b > 1 earlier epochs of the batch are staged at the old length; this epoch keeps that length too —
        // its packets are truncated / zero-padded below — and the new length takes over with the next batch)
      }
      if (L != frame_len) {
        // a different length INSIDE an epoch: the epoch keeps its length — the packet is truncated or zero-padded to it, as
        // the synchronous form does (never dropped: a run of such packets must not stall the epoch, and with it sensing)
        const int n = L < frame_len ? L : frame_len;
        memcpy(pad.data(), pkt, (size_t)n * sizeof(std::complex<float>));
        for (int i = n; i < frame_len; i++) pad[i] = std::complex<float>(0.f, 0.f);
        pkt = pad.data();
      }
      // one copy of the packet into its pinned slot; the epoch's last one also enqueues H2D + kernel + D2H
      const int rc = crn_ingest_push(ring, 0, reinterpret_cast<const float *>(pkt));
      if (rc == CRN_ERR_BUSY) {          // both batch buffers on the GPU: this packet is skipped, like a
        packets_dropped++;               // frame the reference's CE thread was not ready for
        return;
      }
      if (rc != CRN_OK) die_crn();
      if (++fft_counter == packets_in_epoch(frame_len)) {
        ECR->set_ce_sensing(0);  // .cpp:159; the decision is reported by a later execute()
        sensing_on = 0;
        fft_counter = 0;         // .cpp:287-288
      }
      return;
    }
    if (fft_counter == 0) frame_len = L;
    if (L != frame_len) L = L < frame_len ? L : frame_len;  // UHD packet size is constant in practice
    std::complex<float> *dst = &staging[(size_t)fft_counter * frame_len];
    memcpy(dst, pkt, (size_t)L * sizeof(std::complex<float>));
    for (int i = L; i < frame_len; i++) dst[i] = std::complex<float>(0.f, 0.f);
    fft_counter++;

    const int P = packets_in_epoch(frame_len);
    if (fft_counter == P) {  // .cpp:157
      ECR->set_ce_sensing(0);  // .cpp:159
      sensing_on = 0;

      // .cpp:150-154 for all K frames, .cpp:163-197, :200, :214-235, :245-261 — on the GPU
      crn_out out;
      memset(&out, 0, sizeof(out));
      int32_t d = 0;
      float feat[CRN_MAX_BANDS];
      unsigned char occ[CRN_MAX_BANDS];
      double out3[3] = {0.0, 0.0, 0.0};
      out.features = feat;
      out.ann_out = out3;
      out.decision = &d;
      out.occupancy = occ;
      const bool overlapped = cfg.hop != cfg.fft_len;   // Welch: whole frames cut from the run of P packets
      if (crn_sense_run_host(sensor, reinterpret_cast<const float *>(staging.data()), 1, overlapped ? N : frame_len,
                             overlapped ? (int64_t)P * frame_len : 0, &out) != CRN_OK)
        die_crn();
      close_epoch(feat, out3, d, occ);

      fft_counter = 0;  // .cpp:287-288 (fft_avg lives on the device and starts from zero each launch)
    }
  }
}
