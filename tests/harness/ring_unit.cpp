// ring_unit.cpp — TEST INFRASTRUCTURE.  Unit test of the ingest ring's HOST logic (csrc/crn_ingest.cpp compiled against
// tests/harness/fake_hip: no GPU, ThreadSanitizer on): the caller thread and the ring's launcher thread share two batch
// buffers, a work queue and a result queue; this drives every hand-off pattern and checks that each (stream, epoch) comes
// back exactly once, in order per stream, carrying the checksum of exactly its own ten packets.
//   sensing stand-in: features[0] of an epoch = sum of all its samples, features[1] = its first sample, decision = L.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <chrono>
#include <thread>
#include <map>
#include <string>
#include <vector>

#include <hip/hip_runtime.h>   // the stand-in under tests/harness/fake_hip

#include "../../include/crn_sense.h"
#include "../../include/crn_sense_sc16.h"

#include "fake_sense.h"

#define REQUIRE(c)                                                            \
  do {                                                                        \
    if (!(c)) {                                                               \
      fprintf(stderr, "ring_unit: line %d: %s FAILED\n", __LINE__, #c);      \
      exit(1);                                                                \
    }                                                                         \
  } while (0)

struct Feeder {   // deterministic packets: packet p of epoch e of stream s is filled with value(s, e, p)
  int L;
  static float value(int s, long e, int p) { return (float)(s * 1000 + e * 10 + p + 1); }
  std::vector<float> packet(int s, long e, int p) const { return std::vector<float>((size_t)L * 2, value(s, e, p)); }
  float checksum(int s, long e) const {
    double t = 0;
    for (int p = 0; p < 10; p++) t += (double)value(s, e, p) * L * 2;
    return (float)t;
  }
};

static void collect(crn_ingest *g, std::vector<crn_epoch_result> *all) {
  crn_epoch_result r[16];
  int32_t n = 0;
  do {
    REQUIRE(crn_ingest_poll(g, r, 16, &n) == CRN_OK);
    all->insert(all->end(), r, r + n);
  } while (n == 16);
}

static void verify(const std::vector<crn_epoch_result> &all, const Feeder &f, int streams, const std::vector<long> &epochs, long seq0 = 0) {
  std::map<int, long> next;
  size_t want = 0;
  for (int s = 0; s < streams; s++) want += (size_t)epochs[s];
  REQUIRE(all.size() == want);
  for (const crn_epoch_result &r : all) {
    REQUIRE(r.stream >= 0 && r.stream < streams);
    const long e = r.epoch_seq - seq0;
    REQUIRE(e == next[r.stream]);                  // in order per stream, none missing, none twice
    next[r.stream] = e + 1;
    REQUIRE(r.features[0] == f.checksum(r.stream, e));   // exactly its own ten packets
    REQUIRE(r.features[1] == Feeder::value(r.stream, e, 0));
    REQUIRE(r.decision == f.L);
  }
  for (int s = 0; s < streams; s++) REQUIRE(next[s] == epochs[s]);
}

int main() {
  crn_handle h;
  memset(&h, 0, sizeof(h));
  h.cfg.fft_len = h.cfg.hop = 512;
  h.cfg.frames_per_epoch = 10;
  h.cfg.n_bands = 4;
  h.cfg.decide = CRN_DECIDE_ANN;
  crn_ingest *g = NULL;

  // 1. argument checks
  REQUIRE(crn_ingest_create(&h, 0, 364, 1, &g) == CRN_ERR_ARG);
  REQUIRE(crn_ingest_create(&h, 1, 513, 1, &g) == CRN_ERR_ARG);

  // 2. one stream, one epoch per batch (the engine's shape), instant "GPU"
  {
    Feeder f{364};
    REQUIRE(crn_ingest_create(&h, 1, 512, 1, &g) == CRN_OK);
    REQUIRE(crn_ingest_set_packet_len(g, 364) == CRN_OK);
    std::vector<crn_epoch_result> all;
    for (long e = 0; e < 50; e++)
      for (int p = 0; p < 10; p++) {
        std::vector<float> pk = f.packet(0, e, p);
        int rc;
        while ((rc = crn_ingest_push(g, 0, pk.data())) == CRN_ERR_BUSY) collect(g, &all);
        REQUIRE(rc == CRN_OK);
        collect(g, &all);
      }
    REQUIRE(crn_ingest_drain(g) == CRN_OK);
    collect(g, &all);
    verify(all, f, 1, {50});
    crn_ingest_stats st;
    REQUIRE(crn_ingest_get_stats(g, &st) == CRN_OK);
    REQUIRE(st.packets == 500 && st.batches == 50 && st.batches_failed == 0);
    REQUIRE(st.epochs_launched == 50 && st.epochs_ready == 50 && st.epochs_polled == 50);
    REQUIRE(st.latency_us_max > 0 && st.latency_us_sum >= st.latency_us_max && st.latency_us_max < 1e6);
    REQUIRE(crn_ingest_get_stats(g, NULL) == CRN_ERR_ARG);
    REQUIRE(crn_ingest_destroy(g) == CRN_OK);
  }

  // 3. slow "GPU" (2 ms per batch): push must refuse (BUSY), never wait; nothing is lost when the caller retries
  {
    Feeder f{100};
    g_fake_gpu_latency_ns = 2000000;
    REQUIRE(crn_ingest_create(&h, 1, 100, 1, &g) == CRN_OK);
    std::vector<crn_epoch_result> all;
    long busy = 0;
    long long worst = 0;
    for (long e = 0; e < 12; e++)
      for (int p = 0; p < 10; p++) {
        std::vector<float> pk = f.packet(0, e, p);
        for (;;) {
          const long long t0 = fake_hip_now_ns();
          const int rc = crn_ingest_push(g, 0, pk.data());
          const long long dt = fake_hip_now_ns() - t0;
          if (dt > worst) worst = dt;
          if (rc == CRN_OK) break;
          REQUIRE(rc == CRN_ERR_BUSY);
          busy++;
          REQUIRE(crn_ingest_wait(g) == CRN_OK);   // a caller that must not drop waits explicitly
        }
      }
    int64_t dropped = 0;
    REQUIRE(crn_ingest_dropped(g, &dropped) == CRN_OK && dropped == busy && busy > 0);
    REQUIRE(worst < 1500000);                      // no push ever sat out a 2 ms batch
    REQUIRE(crn_ingest_drain(g) == CRN_OK);
    collect(g, &all);
    verify(all, f, 1, {12});
    crn_ingest_stats st;
    REQUIRE(crn_ingest_get_stats(g, &st) == CRN_OK);
    REQUIRE(st.packets == 120 && st.dropped == busy && st.batches == 12 && st.epochs_ready == 12);
    REQUIRE(st.latency_us_sum / (double)st.batches > 1900.0);   // every batch sat out the stand-in's 2 ms
    REQUIRE(crn_ingest_destroy(g) == CRN_OK);
    g_fake_gpu_latency_ns = 0;
  }

  // 4. several streams round-robin, more streams than epochs per batch; then uneven rates (holes carried over); flush
  for (int uneven = 0; uneven < 2; uneven++) {
    Feeder f{64};
    const int S = 5;
    g_fake_gpu_latency_ns = uneven ? 200000 : 0;
    REQUIRE(crn_ingest_create(&h, S, 64, 3, &g) == CRN_OK);
    std::vector<crn_epoch_result> all;
    std::vector<long> done(S, 0);
    std::vector<int> pkt(S, 0);
    const long target[S] = {9, 7, 5, 3, 1};
    bool more = true;
    long round = 0;
    while (more) {
      more = false;
      for (int s = 0; s < S; s++) {
        const int reps = uneven ? (S - s) : 1;     // stream 0 runs five times as fast as stream 4
        for (int k = 0; k < reps && done[s] < target[s]; k++) {
          std::vector<float> pk = f.packet(s, done[s], pkt[s]);
          int rc;
          while ((rc = crn_ingest_push(g, s, pk.data())) == CRN_ERR_BUSY) {
            REQUIRE(crn_ingest_wait(g) == CRN_OK);
            collect(g, &all);
          }
          REQUIRE(rc == CRN_OK);
          if (++pkt[s] == 10) { pkt[s] = 0; done[s]++; }
        }
        if (done[s] < target[s]) more = true;
      }
      if (++round % 7 == 0) REQUIRE(crn_ingest_flush(g) == CRN_OK);   // flushes in the middle of epochs
      collect(g, &all);
    }
    REQUIRE(crn_ingest_drain(g) == CRN_OK);
    collect(g, &all);
    verify(all, f, S, std::vector<long>(target, target + S));
    crn_ingest_stats st;
    REQUIRE(crn_ingest_get_stats(g, &st) == CRN_OK);   // slots whose open epoch moved on to the other buffer carry nothing: not counted
    REQUIRE(st.packets == 250 && st.epochs_launched == 25 && st.epochs_ready == 25 && st.epochs_polled == 25 && st.batches_failed == 0);
    REQUIRE(crn_ingest_destroy(g) == CRN_OK);
  }
  g_fake_gpu_latency_ns = 0;

  // 5. set_packet_len between epochs; refused while an epoch is staged
  {
    REQUIRE(crn_ingest_create(&h, 1, 256, 1, &g) == CRN_OK);
    Feeder f{256};
    std::vector<float> pk = f.packet(0, 0, 0);
    REQUIRE(crn_ingest_push(g, 0, pk.data()) == CRN_OK);
    REQUIRE(crn_ingest_set_packet_len(g, 128) == CRN_ERR_STATE);
    REQUIRE(crn_ingest_set_packet_len(g, 257) == CRN_ERR_ARG);
    for (int p = 1; p < 10; p++) {
      pk = f.packet(0, 0, p);
      REQUIRE(crn_ingest_push(g, 0, pk.data()) == CRN_OK);
    }
    REQUIRE(crn_ingest_drain(g) == CRN_OK);
    REQUIRE(crn_ingest_set_packet_len(g, 128) == CRN_OK);
    Feeder f2{128};
    for (int p = 0; p < 10; p++) {
      pk = f2.packet(0, 1, p);
      REQUIRE(crn_ingest_push(g, 0, pk.data()) == CRN_OK);
    }
    REQUIRE(crn_ingest_drain(g) == CRN_OK);
    std::vector<crn_epoch_result> all;
    collect(g, &all);
    REQUIRE(all.size() == 2 && all[0].decision == 256 && all[1].decision == 128);
    REQUIRE(all[0].features[0] == f.checksum(0, 0) && all[1].features[0] == f2.checksum(0, 1));
    REQUIRE(crn_ingest_destroy(g) == CRN_OK);
  }

  // 6. a launch that fails on the launcher thread: the batch is dropped, the error surfaces once, the ring keeps working
  {
    Feeder f{32};
    REQUIRE(crn_ingest_create(&h, 1, 32, 1, &g) == CRN_OK);
    g_fail_next_launch = 1;
    for (int p = 0; p < 10; p++) {
      std::vector<float> pk = f.packet(0, 0, p);
      REQUIRE(crn_ingest_push(g, 0, pk.data()) == CRN_OK);
    }
    const int rc = crn_ingest_drain(g);
    REQUIRE(rc == CRN_ERR_DEVICE && strstr(crn_last_error(), "forced launch failure") != NULL);
    std::vector<crn_epoch_result> all;
    collect(g, &all);
    REQUIRE(all.empty());
    for (int p = 0; p < 10; p++) {
      std::vector<float> pk = f.packet(0, 1, p);
      REQUIRE(crn_ingest_push(g, 0, pk.data()) == CRN_OK);
    }
    REQUIRE(crn_ingest_drain(g) == CRN_OK);
    collect(g, &all);
    REQUIRE(all.size() == 1 && all[0].epoch_seq == 1 && all[0].features[0] == f.checksum(0, 1));
    crn_ingest_stats st;
    REQUIRE(crn_ingest_get_stats(g, &st) == CRN_OK);
    REQUIRE(st.packets == 20 && st.batches == 1 && st.batches_failed == 1 && st.epochs_launched == 1 && st.epochs_ready == 1);
    REQUIRE(crn_ingest_destroy(g) == CRN_OK);
  }
  // 7. a wire-format ring: int16 packets, the wire-format launch, the other kind of push refused
  {
    REQUIRE(crn_ingest_create_sc16(&h, 2, 100, 2, &g) == CRN_OK);
    std::vector<float> fpk(200, 1.f);
    REQUIRE(crn_ingest_push(g, 0, fpk.data()) == CRN_ERR_STATE);
    std::vector<crn_epoch_result> all;
    long want[2][3] = {{0, 0, 0}, {0, 0, 0}};
    for (int e = 0; e < 3; e++)
      for (int p = 0; p < 10; p++)
        for (int st = 0; st < 2; st++) {
          std::vector<int16_t> pk(200);
          for (int i = 0; i < 200; i++) pk[i] = (int16_t)((st * 7919 + e * 131 + p * 17 + i * 3) % 2001 - 1000);
          for (int i = 0; i < 200; i++) want[st][e] += pk[i];
          int rc;
          while ((rc = crn_ingest_push_sc16(g, st, pk.data())) == CRN_ERR_BUSY) REQUIRE(crn_ingest_wait(g) == CRN_OK);
          REQUIRE(rc == CRN_OK);
          collect(g, &all);
        }
    REQUIRE(crn_ingest_drain(g) == CRN_OK);
    collect(g, &all);
    REQUIRE(all.size() == 6);
    for (const crn_epoch_result &r : all) {
      REQUIRE(r.decision == -100);                                    // the wire-format launch ran, with L = 100
      REQUIRE(r.features[0] == (float)want[r.stream][r.epoch_seq]);   // exactly its own ten packets
    }
    REQUIRE(crn_ingest_destroy(g) == CRN_OK);
    REQUIRE(crn_ingest_create(&h, 1, 100, 1, &g) == CRN_OK);
    std::vector<int16_t> ipk(200, 1);
    REQUIRE(crn_ingest_push_sc16(g, 0, ipk.data()) == CRN_ERR_STATE);
    REQUIRE(crn_ingest_destroy(g) == CRN_OK);
  }
  // 8. random schedules (seeded): any number of streams and epochs per batch, streams picked at random (so their epochs
  // interleave arbitrarily), flushes at random moments, a "GPU" of random latency, callers that wait or retry on BUSY — every
  // complete epoch exactly once, in order per stream, with exactly its own packets; epochs still open at the end never appear
  for (unsigned seed = 1; seed <= 12; seed++) {
    srand(seed);
    const int S = 1 + rand() % 6, B = 1 + rand() % 5, L = 8 + rand() % 57;
    g_fake_gpu_latency_ns = (rand() % 3 == 0) ? 0 : 20000 + rand() % 300000;
    Feeder f{L};
    REQUIRE(crn_ingest_create(&h, S, L, B, &g) == CRN_OK);
    std::vector<long> done(S, 0);
    std::vector<int> pkt(S, 0);
    std::vector<crn_epoch_result> all;
    const int steps = 400 + rand() % 1200;
    for (int it = 0; it < steps; it++) {
      const int st = rand() % S;
      std::vector<float> pk = f.packet(st, done[st], pkt[st]);
      int rc;
      while ((rc = crn_ingest_push(g, st, pk.data())) == CRN_ERR_BUSY) {
        if (rand() % 2) REQUIRE(crn_ingest_wait(g) == CRN_OK);
        collect(g, &all);
      }
      REQUIRE(rc == CRN_OK);
      if (++pkt[st] == 10) { pkt[st] = 0; done[st]++; }
      if (rand() % 50 == 0) REQUIRE(crn_ingest_flush(g) == CRN_OK);
      if (rand() % 7 == 0) collect(g, &all);
    }
    REQUIRE(crn_ingest_drain(g) == CRN_OK);
    collect(g, &all);
    verify(all, f, S, done);
    crn_ingest_stats st;
    REQUIRE(crn_ingest_get_stats(g, &st) == CRN_OK);
    long total = 0;
    for (int i = 0; i < S; i++) total += done[i];
    REQUIRE(st.packets == steps && st.epochs_ready == total && st.epochs_polled == total && st.batches_failed == 0);
    REQUIRE(crn_ingest_destroy(g) == CRN_OK);
  }
  g_fake_gpu_latency_ns = 0;

  // 9. overlapped frames (a Welch plan: hop = N/2): an epoch is ONE contiguous run of P = ceil(((K - 1) hop + N) / L) packets, the
  // launch asks for whole frames (samples_per_frame = N) with the epoch stride P L; a change of packet length changes P; three
  // streams interleaved; the stand-in's checksum covers exactly the (K - 1) hop + N samples the kernel would read
  {
    crn_handle hw = h;
    hw.cfg.fft_len = 1024;
    hw.cfg.hop = 512;
    hw.cfg.frames_per_epoch = 8;
    hw.cfg.decide = CRN_DECIDE_THRESHOLD;
    const long span = 7 * 512 + 1024;
    for (int L : {364, 512, 1000}) {
      const int S = 3, P = (int)((span + L - 1) / L);
      REQUIRE(crn_ingest_create(&hw, S, 1024, 2, &g) == CRN_OK);
      REQUIRE(crn_ingest_set_packet_len(g, L) == CRN_OK);
      int32_t pp = 0;
      REQUIRE(crn_ingest_packets_per_epoch(g, &pp) == CRN_OK && pp == P);
      std::vector<crn_epoch_result> all;
      std::vector<double> want((size_t)S * 4, 0.0);
      for (long e = 0; e < 4; e++)
        for (int p = 0; p < P; p++)
          for (int st = 0; st < S; st++) {
            std::vector<float> pk((size_t)L * 2);
            for (int i = 0; i < L; i++) {
              pk[2 * i] = (float)((st * 31 + e * 7 + p * 3 + i) % 97);
              pk[2 * i + 1] = 0.5f;
              if ((long)p * L + i < span) want[(size_t)st * 4 + e] += (double)pk[2 * i] + 0.5;   // the tail of the last packet is beyond the last frame
            }
            int rc;
            while ((rc = crn_ingest_push(g, st, pk.data())) == CRN_ERR_BUSY) REQUIRE(crn_ingest_wait(g) == CRN_OK);
            REQUIRE(rc == CRN_OK);
            collect(g, &all);
          }
      REQUIRE(crn_ingest_drain(g) == CRN_OK);
      collect(g, &all);
      REQUIRE(all.size() == (size_t)S * 4);
      REQUIRE(g_fake_last_L.load() == 1024 && g_fake_last_stride.load() == (long long)P * L);
      for (const crn_epoch_result &r : all) REQUIRE(r.features[0] == (float)want[(size_t)r.stream * 4 + r.epoch_seq]);
      REQUIRE(crn_ingest_destroy(g) == CRN_OK);
    }
  }

  // 10. crn_ingest_wait waits for the buffer that blocks the push: with more streams than a hand-off can complete, a push is
  // refused because the FULL fill buffer has open epochs that must move to the other buffer, which is still on the "GPU" — a
  // caller that waits and pushes again must get through on the first retry (it used to spin: the fill buffer itself was free)
  {
    g_fake_gpu_latency_ns = 3000000;
    Feeder f{16};
    const int S = 3;
    REQUIRE(crn_ingest_create(&h, S, 16, 1, &g) == CRN_OK);   // three slots per buffer, a launch per complete epoch
    std::vector<crn_epoch_result> all;
    std::vector<long> done(S, 0);
    std::vector<int> pkt(S, 0);
    long refusals = 0, worst_retries = 0;
    // stream 0 runs ten times as fast as the others, so its epochs complete while theirs stay open (holes to carry over)
    for (int it = 0; it < 600; it++) {
      const int st = it % 12 < 10 ? 0 : 1 + (it / 12) % 2;
      std::vector<float> pk = f.packet(st, done[st], pkt[st]);
      long retries = 0;
      int rc;
      while ((rc = crn_ingest_push(g, st, pk.data())) == CRN_ERR_BUSY) {
        refusals++;
        retries++;
        REQUIRE(crn_ingest_wait(g) == CRN_OK);
        collect(g, &all);
      }
      REQUIRE(rc == CRN_OK);
      if (retries > worst_retries) worst_retries = retries;
      if (++pkt[st] == 10) { pkt[st] = 0; done[st]++; }
    }
    REQUIRE(refusals > 0 && worst_retries == 1);   // one wait is enough, whichever buffer was in the way
    REQUIRE(crn_ingest_drain(g) == CRN_OK);
    collect(g, &all);
    verify(all, f, S, done);
    REQUIRE(crn_ingest_destroy(g) == CRN_OK);
    g_fake_gpu_latency_ns = 0;
  }
  // noise-floor calibration on the launcher thread (crn_ingest_calibrate): posted by this thread without a HIP call, carried out
  // by the launcher after the n-th epoch came back; every epoch launched before the thresholds landed is marked, the first
  // unmarked one carries the estimate; a slow "GPU" makes batches overlap the calibration
  for (int slow = 0; slow < 2; slow++) {
    crn_handle ht;
    memset(&ht, 0, sizeof(ht));
    ht.cfg.fft_len = ht.cfg.hop = 512;
    ht.cfg.frames_per_epoch = 10;
    ht.cfg.n_bands = 8;
    ht.cfg.decide = CRN_DECIDE_THRESHOLD;
    g_fake_gpu_latency_ns = slow ? 1000000 : 0;
    REQUIRE(crn_ingest_calibrate(g = NULL, 4, 2.f) == CRN_ERR_ARG);
    const int reserved0 = g_fake_nf_reserved.load(), updates0 = g_fake_threshold_updates.load();
    REQUIRE(crn_ingest_create(&ht, 1, 64, 1, &g) == CRN_OK);
    REQUIRE(g_fake_nf_reserved.load() == reserved0 + 1);                      // buffers made at creation, not by the request
    REQUIRE(crn_ingest_calibrate(g, 0, 2.f) == CRN_ERR_ARG && crn_ingest_calibrate(g, 4, 0.f) == CRN_ERR_ARG);
    float nf = -1.f;
    int32_t busy = -1;
    REQUIRE(crn_ingest_noise_floor(g, &nf, &busy) == CRN_OK && nf == 0.f && busy == 0);
    fake_hip_watch_thread = true;
    const long long calls0 = fake_hip_calls_on_watched_threads.load();
    REQUIRE(crn_ingest_calibrate(g, 4, 2.f) == CRN_OK);
    REQUIRE(fake_hip_calls_on_watched_threads.load() == calls0);               // only a request: no HIP call on the posting thread
    fake_hip_watch_thread = false;
    REQUIRE(crn_ingest_calibrate(g, 4, 2.f) == CRN_ERR_STATE);                 // one at a time
    std::vector<crn_epoch_result> all;
    for (long e = 0; e < 12; e++)
      for (int p = 0; p < 10; p++) {
        std::vector<float> pk((size_t)64 * 2, (float)(e + 1));                 // features[1] of epoch e = e + 1
        int rc;
        while ((rc = crn_ingest_push(g, 0, pk.data())) == CRN_ERR_BUSY) { REQUIRE(crn_ingest_wait(g) == CRN_OK); collect(g, &all); }
        REQUIRE(rc == CRN_OK);
      }
    REQUIRE(crn_ingest_drain(g) == CRN_OK);
    collect(g, &all);
    REQUIRE(all.size() == 12 && g_fake_threshold_updates.load() == updates0 + 1);
    REQUIRE(crn_ingest_noise_floor(g, &nf, &busy) == CRN_OK && nf == 2.5f && busy == 0);   // mean of 1, 2, 3, 4
    REQUIRE(g_fake_thresholds_set[0] == 5.0f && g_fake_thresholds_set[7] == 5.0f && ht.cfg.thresh[3] == 5.0f);
    size_t marked = 0;
    while (marked < all.size() && (all[marked].flags & CRN_EPOCH_CALIBRATION)) marked++;
    REQUIRE(marked >= 4 && marked < 12);                                         // the four that fed it (+ any launched before it landed)
    if (!slow) REQUIRE(marked <= 5);
    for (size_t i = 0; i < all.size(); i++) {
      REQUIRE(all[i].epoch_seq == (long)i);
      REQUIRE(((all[i].flags & CRN_EPOCH_CALIBRATION) != 0) == (i < marked));    // marked epochs first, none after the update
      REQUIRE(all[i].noise_floor == (i < marked ? 0.f : 2.5f));
    }
    REQUIRE(crn_ingest_calibrate(g, 2, 3.f) == CRN_OK);                          // again later: re-calibration
    REQUIRE(crn_ingest_destroy(g) == CRN_OK);
    // a handle that decides by the network has nothing to calibrate
    REQUIRE(crn_ingest_create(&h, 1, 64, 1, &g) == CRN_OK);
    REQUIRE(crn_ingest_calibrate(g, 4, 2.f) == CRN_ERR_STATE);
    REQUIRE(crn_ingest_destroy(g) == CRN_OK);
    g_fake_gpu_latency_ns = 0;
  }
  // 12. pre-wake: ten packets before the hand-off of a small batch the pushing thread tells the launcher, which then polls for the
  //     work instead of sleeping (bounded by $CRN_INGEST_PREWAKE_US).  Packets a radio's interval apart, epochs a sensing period
  //     apart; the ring destroyed while the launcher is still polling for a hand-off that never comes; and switched off.
  for (const char *us : {"600", "20000", "0"}) {
    setenv("CRN_INGEST_PREWAKE_US", us, 1);
    for (int B : {1, 3}) {
      const long warm_before = g_fake_warm_launches.load();
      Feeder f{364};
      REQUIRE(crn_ingest_create(&h, 1, 364, B, &g) == CRN_OK);
      std::vector<crn_epoch_result> all;
      for (long e = 0; e < 6; e++) {
        for (int p = 0; p < 10; p++) {
          std::vector<float> pk = f.packet(0, e, p);
          REQUIRE(crn_ingest_push(g, 0, pk.data()) == CRN_OK);
          std::this_thread::sleep_for(std::chrono::microseconds(28));
          collect(g, &all);
        }
        std::this_thread::sleep_for(std::chrono::milliseconds(3));
      }
      REQUIRE(crn_ingest_drain(g) == CRN_OK);
      collect(g, &all);
      verify(all, f, 1, {6});
      // a pre-wake that follows an idle stretch (> 2 ms since the last launch) also queues one empty launch on the ring's stream
      const long warm = g_fake_warm_launches.load() - warm_before;
      REQUIRE(atoi(us) == 0 ? warm == 0 : (warm >= 1 && warm <= 6));
      // an epoch that stays open: the launcher gives up polling after its budget; destroy never waits longer than that
      std::vector<float> pk = f.packet(0, 6, 0);
      REQUIRE(crn_ingest_push(g, 0, pk.data()) == CRN_OK);
      const auto t0 = std::chrono::steady_clock::now();
      REQUIRE(crn_ingest_destroy(g) == CRN_OK);
      REQUIRE(std::chrono::steady_clock::now() - t0 < std::chrono::milliseconds(500));
    }
  }
  unsetenv("CRN_INGEST_PREWAKE_US");
  REQUIRE(g_fake_rings_attached.load() == 0);   // every ring that attached to its handle detached again
  printf("ring_unit: ok\n");
  return 0;
}
