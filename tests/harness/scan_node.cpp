// scan_node.cpp — TEST INFRASTRUCTURE / example.  BASELINE.json configs[4] (wideband scan: streams of 64 channels
// sharded over the node's GPUs, RCCL gather of the occupancy vector) written as a C++ host program over the C ABI
// alone — no Python, no torch: what a CRTS-style C++ node does with include/crn_sense.h.
//
//   one process per GPU:   RANK=r WORLD_SIZE=w LOCAL_RANK=l  scan_node <streams_total> <epochs_per_stream> <steps> <id_file>
// rank 0 creates the RCCL unique id through crn_comm_unique_id and publishes it in <id_file>; the other ranks read it
// there (any out-of-band channel does).  Every rank owns a contiguous block of streams (weak scaling), generates their
// traffic on its device (the reference's Markov primary-user model), and per step runs the sensing kernel into a slot of
// the communicator and queues the all-gather on the side stream; at the end it checks that its own block sits unchanged at
// its place in the gathered vector and prints one line.
// crn_comm_create is collective and — like ncclCommInitRank under it — cannot tell the other ranks about a local failure: a rank
// that fails before it would leave the others waiting inside the collective.  So the ranks first AGREE, over the same out-of-band
// channel that carries the id (files here), that every one of them finished its local set-up; if one did not, all of them stop.
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include <chrono>
#include <vector>

#include "crn_sense.h"

#define CHECK(x)                                                     \
  do {                                                               \
    if ((x) != CRN_OK) {                                             \
      fprintf(stderr, "scan_node: %s: %s\n", #x, crn_last_error()); \
      return 1;                                                      \
    }                                                                \
  } while (0)
#define HIP(x)                                                                  \
  do {                                                                          \
    hipError_t e_ = (x);                                                        \
    if (e_ != hipSuccess) {                                                     \
      fprintf(stderr, "scan_node: %s: %s\n", #x, hipGetErrorString(e_));        \
      return 1;                                                                 \
    }                                                                           \
  } while (0)

// Every rank publishes "<id_file>.rank<r>" = 1 (local set-up done) or 0 (failed), then waits for all of them: true only if every
// rank said 1 within the time limit.
static bool all_ranks_ready(const char *id_file, int rank, int world, bool ok) {
  char path[4096], tmp[4200];
  snprintf(path, sizeof(path), "%s.rank%d", id_file, rank);
  snprintf(tmp, sizeof(tmp), "%s.tmp", path);
  FILE *f = fopen(tmp, "w");
  if (!f) return false;
  fputc(ok ? '1' : '0', f);
  fclose(f);
  if (rename(tmp, path) != 0) return false;
  bool all = ok;
  for (int r = 0; r < world && all; r++) {
    snprintf(path, sizeof(path), "%s.rank%d", id_file, r);
    int c = EOF;
    for (int tries = 0; tries < 1200 && c == EOF; tries++) {   // up to 2 minutes per rank
      if ((f = fopen(path, "r")) != NULL) {
        c = fgetc(f);
        fclose(f);
      }
      if (c == EOF) usleep(100000);
    }
    all = c == '1';
    if (!all) fprintf(stderr, "scan_node: rank %d: rank %d %s: stopping before the collective\n", rank, r, c == '0' ? "failed its set-up" : "never reported");
  }
  return all;
}

static int env_int(const char *k, int d) {
  const char *v = getenv(k);
  return v ? atoi(v) : d;
}

int main(int argc, char **argv) {
  if (argc < 5) {
    fprintf(stderr, "usage: RANK= WORLD_SIZE= LOCAL_RANK= %s streams_total epochs_per_stream steps id_file\n", argv[0]);
    return 2;
  }
  const int rank = env_int("RANK", 0), world = env_int("WORLD_SIZE", 1), device = env_int("LOCAL_RANK", 0);
  const int streams_total = atoi(argv[1]), eps = atoi(argv[2]), steps = atoi(argv[3]);
  const char *id_file = argv[4];
  // contiguous block of streams for this rank (remainder to the low ranks)
  const int base = streams_total / world, rem = streams_total % world;
  const int lo = rank * base + (rank < rem ? rank : rem), n_streams = base + (rank < rem ? 1 : 0);
  if (streams_total % world != 0) {
    fprintf(stderr, "scan_node: streams_total must be a multiple of WORLD_SIZE (equal blocks for the all-gather)\n");
    return 2;
  }
  const int64_t E = (int64_t)n_streams * eps;

  crn_cfg cfg;
  crn_handle *h = NULL;
  hipStream_t stream = NULL;
  float *d_iq = NULL, *d_feat = NULL;
  int32_t *d_truth = NULL, *d_dec = NULL;
  int64_t spe = 0;
  uint8_t id[CRN_COMM_ID_BYTES];
  // everything a rank does on its own, before the first collective
  auto local_setup = [&]() -> int {
  CHECK(crn_cfg_welch(&cfg, 4096, 8, 64));
  // per-band threshold = 4 x the noise floor, measured below on this node's own streams (NF_est = the
  // median band energy); until then nothing is flagged
  for (int b = 0; b < 64; b++) cfg.thresh[b] = INFINITY;
  cfg.device = device;
  CHECK(crn_sense_create(&cfg, &h));
  HIP(hipSetDevice(device));
  HIP(hipStreamCreate(&stream));

  spe = (int64_t)cfg.frames_per_epoch * cfg.hop;            // dense epochs, hop N/2
  const int64_t n_samples = E * spe + (cfg.fft_len - cfg.hop);
  HIP(hipMalloc((void **)&d_iq, (size_t)n_samples * 8));
  HIP(hipMemset(d_iq, 0, (size_t)n_samples * 8));
  HIP(hipMalloc((void **)&d_truth, (size_t)E * 4));
  HIP(hipMalloc((void **)&d_dec, (size_t)E * 4));
  HIP(hipMalloc((void **)&d_feat, (size_t)E * 64 * 4));
  crn_synth_cfg sc;
  memset(&sc, 0, sizeof(sc));
  sc.seed = 0xC0FFEEull + 1000ull * (uint64_t)rank;
  sc.noise_power = 1e-6f;
  sc.signal_rms = 0.02f;
  sc.tones_per_band = 8;
  sc.pu_model = CRN_PU_MARKOV_INTENDED;   // CE_PU_MARKOV_Chain_Tx.cpp:88-128 as intended, one chain per stream
  sc.signal_kind = CRN_SIG_TONES;
  sc.n_streams = n_streams;
  CHECK(crn_synth_fill_device_ex(h, &sc, d_iq, E, spe, d_truth, stream));

  {  // calibrate: one pass for the features, the median band energy, thresholds = 4 x it (64 x 4096 x 1e-6 x 3/8 = 0.098 expected)
    crn_out o = {d_feat, NULL, NULL, NULL, NULL};
    CHECK(crn_sense_run_device(h, d_iq, E, cfg.fft_len, 0, &o, stream));
    float nf = 0.f, thr[64];
    CHECK(crn_noise_floor_device(h, d_feat, E, &nf, stream));
    if (!(nf > 0.08f && nf < 0.13f)) {
      fprintf(stderr, "scan_node: rank %d: noise-floor estimate %g is off\n", rank, nf);
      return 1;
    }
    for (int b = 0; b < 64; b++) thr[b] = 4.0f * nf;
    CHECK(crn_sense_set_thresholds(h, thr, 64, stream));
    if (rank == 0) printf("scan_node: noise floor (median band energy) %.5f -> thresholds %.5f\n", nf, 4.0f * nf);
  }

  // the RCCL unique id: rank 0 makes it, the others pick it up from the file
  if (rank == 0) {
    CHECK(crn_comm_unique_id(id));
    char tmp[4096];
    snprintf(tmp, sizeof(tmp), "%s.tmp", id_file);
    FILE *f = fopen(tmp, "wb");
    if (!f || fwrite(id, 1, sizeof(id), f) != sizeof(id)) return 1;
    fclose(f);
    rename(tmp, id_file);
  } else {
    FILE *f = NULL;
    for (int tries = 0; tries < 600 && !(f = fopen(id_file, "rb")); tries++) usleep(100000);
    if (!f || fread(id, 1, sizeof(id), f) != sizeof(id)) {
      fprintf(stderr, "scan_node: rank %d: no unique id in %s\n", rank, id_file);
      return 1;
    }
    fclose(f);
  }
  return 0;
  };
  const int setup_rc = local_setup();
  if (!all_ranks_ready(id_file, rank, world, setup_rc == 0)) return 1;
  crn_comm *comm = NULL;
  CHECK(crn_comm_create(device, rank, world, id, E * 64, 2, &comm));
  {  // what RCCL itself says the communicator is: it must be this job's `world` ranks, and this rank's place in it
    crn_comm_info_t ci;
    CHECK(crn_comm_info(comm, &ci));
    if (ci.nranks != world || ci.rank != rank) {
      fprintf(stderr, "scan_node: rank %d/%d: RCCL reports a communicator of %d ranks, this one rank %d\n", rank, world, ci.nranks, ci.rank);
      return 1;
    }
    if (rank == 0) printf("scan_node: RCCL communicator: %d ranks, rank 0 on device %d, version %d, library %s\n", ci.nranks, ci.rccl_device, ci.rccl_version, ci.library);
  }

  auto step = [&](int64_t i) -> int {
    uint8_t *occ = NULL;
    CHECK(crn_comm_local(comm, i, stream, &occ));
    crn_out o = {d_feat, NULL, d_dec, occ, NULL};
    CHECK(crn_sense_run_device(h, d_iq, E, cfg.fft_len, 0, &o, stream));
    CHECK(crn_comm_allgather(comm, i, stream));
    return 0;
  };
  for (int i = 0; i < 3; i++)
    if (step(i)) return 1;
  CHECK(crn_comm_finish(comm, stream));
  HIP(hipStreamSynchronize(stream));
  const auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < steps; i++)
    if (step(3 + i)) return 1;
  CHECK(crn_comm_finish(comm, stream));
  HIP(hipStreamSynchronize(stream));
  const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();

  // every rank must find its own block, unchanged, at its place in the gathered vector
  const int64_t last = 3 + steps - 1;
  uint8_t *d_local = NULL;
  const uint8_t *d_all = NULL;
  CHECK(crn_comm_local(comm, last, stream, &d_local));
  CHECK(crn_comm_gathered(comm, last, &d_all));
  std::vector<uint8_t> mine((size_t)E * 64), all((size_t)world * E * 64);
  std::vector<int32_t> truth((size_t)E);
  HIP(hipMemcpy(mine.data(), d_local, mine.size(), hipMemcpyDeviceToHost));
  HIP(hipMemcpy(all.data(), d_all, all.size(), hipMemcpyDeviceToHost));
  HIP(hipMemcpy(truth.data(), d_truth, truth.size() * 4, hipMemcpyDeviceToHost));
  const bool placed = memcmp(all.data() + (size_t)rank * E * 64, mine.data(), mine.size()) == 0;
  // the driven channel of every epoch (Markov truth: band 1..3 -> channel index truth - 1 of the 64) must read occupied
  int64_t hit = 0, occupied = 0;
  for (int64_t e = 0; e < E; e++) {
    hit += mine[(size_t)e * 64 + (size_t)(truth[e] - 1)] != 0;
    for (int b = 0; b < 64; b++) occupied += mine[(size_t)e * 64 + b];
  }
  printf("scan_node rank %d/%d device %d: streams [%d, %d) x %d epochs x 64 channels, %d steps: %.1f Msamples/s on this rank, "
         "gathered %lld x 64 occupancy bytes, own block in place: %s, driven channel flagged in %lld of %lld epochs, %.2f channels occupied per epoch\n",
         rank, world, device, lo, lo + n_streams, eps, steps, (double)E * spe * steps / dt / 1e6, (long long)((int64_t)world * E),
         placed ? "yes" : "NO", (long long)hit, (long long)E, (double)occupied / (double)E);
  crn_comm_destroy(comm);
  crn_sense_destroy(h);
  // the driven channel every epoch; a few more may read occupied (an epoch's last frame reaches half a frame into the
  // next epoch's traffic, and a tone that starts mid-frame splatters across bands)
  return placed && hit == E && occupied < 8 * E ? 0 : 1;
}
