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
  packets_dropped = 0;
  memset(features, 0, sizeof(features));
  memset(outputs, 0, sizeof(outputs));

  crn_cfg_reference(&cfg);

  // ce_args from the scenario file arrive as argv (reference: src/crts.cpp:43-81 str2argcargv,
  // which resets optind; CE_Template.cpp:17-25 shows the getopt idiom)
  int o;
  optind = 1;
  stats_on = 0;
  while ((o = getopt(argc, argv, "a:d:g:s:v:")) != EOF) {
    switch (o) {
    case 'a': async_mode = atoi(optarg); break;            // 0: decide inside the K-th execute()
    case 'd': cfg.device = atoi(optarg); break;           // HIP device ordinal
    case 'g': wall_clock_gate = atoi(optarg) != 0; break;  // 0: sense continuously
    case 's': stats_on = atoi(optarg); break;               // 1: time every launch, one summary line at release()
    case 'v': verbose = atoi(optarg); break;
    }
  }

  struct timeval tv;
  gettimeofday(&tv, NULL);
  sense_time_s = tv.tv_sec;
  sense_time_us = tv.tv_usec;

  // replaces memset of the three buffers + fft_create_plan (.cpp:36-45)
  if (crn_sense_create(&cfg, &sensor) != CRN_OK) die_crn();
  if (crn_sense_reserve_host(sensor, 1, 0) != CRN_OK) die_crn();  // scratch + pinned staging + kernel load
  if (stats_on && crn_sense_set_timing(sensor, 1) != CRN_OK) die_crn();
  if (async_mode) {
    // one stream, one epoch per batch; sized for full-length packets, the actual UHD packet length is
    // known only when the rx worker starts (src/extensible_cognitive_radio.cpp:1263-1265)
    if (crn_ingest_create(sensor, 1, cfg.fft_len, 1, &ring) != CRN_OK) die_crn();
    frame_len = cfg.fft_len;
  } else {
    staging.assign((size_t)cfg.frames_per_epoch * cfg.fft_len, std::complex<float>(0.f, 0.f));
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

// The reference's report + action block (.cpp:202-261) for one closed epoch.
void CE_Predictive_Node_GPU::report(const float *feat, const double *out3, int d) {
  memcpy(features, feat, sizeof(features));
  memcpy(outputs, out3, sizeof(outputs));
  decision = d;
  epochs_closed++;
  if (verbose) {  // .cpp:202-207, 239-241
    printf("--------------------------------------------------------------\n");
    printf("-            		FEATURES BUFFER 	               -\n");
    printf("--------------------------------------------------------------\n");
    printf("NOISE FLOOR   %.2e\nCH1           %.2e\nCH2           %.2e\nCH3           %.2e\n ", features[0],
           features[1], features[2], features[3]);
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

  // a decision launched by an earlier call may have landed (one event query, never a wait)
  if (ring) {
    crn_epoch_result r;
    int32_t n = 0;
    while (crn_ingest_poll(ring, &r, 1, &n) == CRN_OK && n == 1) report(r.features, r.ann_out, r.decision);
  }

  // handle samples (.cpp:146)
  if (ECR->CE_metrics.CE_event == ExtensibleCognitiveRadio::USRP_RX_SAMPS) {
    const int N = cfg.fft_len, K = cfg.frames_per_epoch;
    // .cpp:149 copies ce_usrp_rx_buffer_length samples unchecked; a packet longer than the FFT
    // would overrun the reference's buffer — here it is truncated to N.
    int L = ECR->ce_usrp_rx_buffer_length;
    if (L > N) L = N;
    if (L < 1) return;
    if (ring) {
      if (L != frame_len) {              // UHD packet size is constant in practice: once, at the first packet
        if (fft_counter != 0) return;    // never inside an epoch
        const int rc = crn_ingest_set_packet_len(ring, L);
        if (rc == CRN_ERR_BUSY) { packets_dropped++; return; }
        if (rc != CRN_OK) die_crn();
        frame_len = L;
      }
      // one copy of the packet into its pinned slot; the K-th one also enqueues H2D + kernel + D2H
      const int rc = crn_ingest_push(ring, 0, reinterpret_cast<const float *>(ECR->ce_usrp_rx_buffer));
      if (rc == CRN_ERR_BUSY) {          // both batch buffers on the GPU: this packet is skipped, like a
        packets_dropped++;               // frame the reference's CE thread was not ready for
        return;
      }
      if (rc != CRN_OK) die_crn();
      if (++fft_counter == K) {
        ECR->set_ce_sensing(0);  // .cpp:159; the decision is reported by a later execute()
        sensing_on = 0;
        fft_counter = 0;         // .cpp:287-288
      }
      return;
    }
    if (fft_counter == 0) frame_len = L;
    if (L != frame_len) L = L < frame_len ? L : frame_len;  // UHD packet size is constant in practice
    std::complex<float> *dst = &staging[(size_t)fft_counter * frame_len];
    memcpy(dst, ECR->ce_usrp_rx_buffer, (size_t)L * sizeof(std::complex<float>));
    for (int i = L; i < frame_len; i++) dst[i] = std::complex<float>(0.f, 0.f);
    fft_counter++;

    if (fft_counter == K) {  // .cpp:157
      ECR->set_ce_sensing(0);  // .cpp:159
      sensing_on = 0;

      // .cpp:150-154 for all K frames, .cpp:163-197, :200, :214-235, :245-261 — on the GPU
      crn_out out;
      memset(&out, 0, sizeof(out));
      int32_t d = 0;
      float feat[4];
      double out3[3];
      out.features = feat;
      out.ann_out = out3;
      out.decision = &d;
      if (crn_sense_run_host(sensor, reinterpret_cast<const float *>(staging.data()), 1, frame_len, 0, &out) !=
          CRN_OK)
        die_crn();
      report(feat, out3, d);

      fft_counter = 0;  // .cpp:287-288 (fft_avg lives on the device and starts from zero each launch)
    }
  }
}
