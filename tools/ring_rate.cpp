// ring_rate.cpp — how many radios one host thread can feed: S streams of reference-sized packets (364 samples) pushed round-robin
// into one ingest ring (crn_ingest_*, C ABI only), decisions polled as they come.  Prints the sustained rate, what was refused and
// the ring's own latency counters.  Build: make -C tests/harness ring_rate ; run on a GPU box: tools/ring_rate [streams] [epochs_per_batch] [seconds] [sc16]
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <random>
#include <vector>

#include "../include/crn_sense.h"
#ifdef CRN_WITH_SC16
#include "../include/crn_sense_sc16.h"
#endif

#define CHECK(x)                                                       \
  do {                                                                 \
    if ((x) != CRN_OK) {                                               \
      fprintf(stderr, "%s: %s\n", #x, crn_last_error());               \
      return 1;                                                        \
    }                                                                  \
  } while (0)

int main(int argc, char **argv) {
  const int S = argc > 1 ? atoi(argv[1]) : 64;
  const int B = argc > 2 ? atoi(argv[2]) : 256;
  const double seconds = argc > 3 ? atof(argv[3]) : 3.0;
  // packets in the radio's wire format (int16 pairs): the optional entry points — build with -DCRN_WITH_SC16 against libcrnsense_sc16.so
#ifdef CRN_WITH_SC16
  const bool sc16 = argc > 4 && strcmp(argv[4], "sc16") == 0;
#else
  const bool sc16 = false;
  if (argc > 4 && strcmp(argv[4], "sc16") == 0) { fprintf(stderr, "ring_rate: built without CRN_WITH_SC16\n"); return 2; }
#endif
  const int L = 364;
  crn_cfg cfg;
  CHECK(crn_cfg_reference(&cfg));
  crn_handle *h = NULL;
  CHECK(crn_sense_create(&cfg, &h));
  CHECK(crn_sense_set_timing(h, 1));
  crn_ingest *g = NULL;
#ifdef CRN_WITH_SC16
  CHECK(sc16 ? crn_ingest_create_sc16(h, S, L, B, &g) : crn_ingest_create(h, S, L, B, &g));
#else
  CHECK(crn_ingest_create(h, S, L, B, &g));
#endif
  // a few MB of packet data, walked cyclically (so that the source is not one cache-resident packet)
  const int n_src = 4096;
  std::vector<float> src((size_t)n_src * L * 2);
  std::mt19937 rng(1);
  std::normal_distribution<float> nd(0.f, 1e-3f);
  for (float &v : src) v = nd(rng);
  std::vector<int16_t> src16(src.size());
  for (size_t i = 0; i < src.size(); i++) src16[i] = (int16_t)(src[i] * 32768.f);
  std::vector<crn_epoch_result> res(1024);
  long long pushed = 0, refused = 0, decisions = 0;
  int32_t n = 0;
  const auto t0 = std::chrono::steady_clock::now();
  double elapsed = 0;
  for (long long it = 0;; it++) {
    for (int s = 0; s < S; s++) {
      const size_t at = (size_t)((it * S + s) % n_src) * L * 2;
#ifdef CRN_WITH_SC16
      const int rc = sc16 ? crn_ingest_push_sc16(g, s, src16.data() + at) : crn_ingest_push(g, s, src.data() + at);
#else
      const int rc = crn_ingest_push(g, s, src.data() + at);
#endif
      if (rc == CRN_OK) pushed++;
      else if (rc == CRN_ERR_BUSY) refused++;
      else { fprintf(stderr, "push: %s\n", crn_last_error()); return 1; }
    }
    CHECK(crn_ingest_poll(g, res.data(), (int32_t)res.size(), &n));
    decisions += n;
    if ((it & 63) == 0) {
      elapsed = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      if (elapsed >= seconds) break;
    }
  }
  CHECK(crn_ingest_drain(g));
  do {
    CHECK(crn_ingest_poll(g, res.data(), (int32_t)res.size(), &n));
    decisions += n;
  } while (n > 0);
  crn_ingest_stats st;
  crn_sense_stats ss;
  CHECK(crn_ingest_get_stats(g, &st));
  CHECK(crn_sense_get_stats(h, &ss));
  printf("ring_rate: %d streams x %d-sample packets%s, %d epochs per batch, one pushing thread, %.2f s\n", S, L,
         sc16 ? " in wire format (int16 pairs)" : "", B, elapsed);
  printf("  accepted %.1f Msamples/s (%.2f GB/s of IQ) = %.1f radios at 13 Msamples/s; refused %.2f %% of the packets offered\n",
         pushed * (double)L / elapsed / 1e6, pushed * (double)L * (sc16 ? 4 : 8) / elapsed / 1e9, pushed * (double)L / elapsed / 13e6,
         100.0 * refused / (double)(pushed + refused));
  printf("  decisions %lld (%.0f /s); batches %lld, hand-off to results %.0f us mean / %.0f us max; kernel %.1f us mean per batch\n", decisions,
         decisions / elapsed, (long long)st.batches, st.batches ? st.latency_us_sum / st.batches : 0.0, st.latency_us_max,
         ss.timed_launches ? 1e3 * ss.kernel_ms / ss.timed_launches : 0.0);
  crn_ingest_destroy(g);
  crn_sense_destroy(h);
  return 0;
}
