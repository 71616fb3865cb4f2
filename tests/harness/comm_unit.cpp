// comm_unit.cpp — TEST INFRASTRUCTURE.  The occupancy exchange (csrc/crn_comm.cpp) as a world of TWO ranks without a GPU:
// two threads, each with its own communicator, the HIP calls replaced by host stand-ins defined here (link-time: libamdhip64 is
// not linked) and RCCL by tests/harness/libfake_rccl.so ($CRN_RCCL_LIB); worlds of 1, 2, 3, 5 and 8 ranks x 1, 2, 3 slots.  Checks what the unattended 8-GPU run relies on:
// every rank receives [rank 0 block][rank 1 block], slot i % depth, slots reused after `depth` steps, sizes and offsets right.
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <thread>
#include <vector>

#include "../../include/crn_sense.h"

// ---- host stand-ins for the HIP runtime calls crn_comm.cpp makes ("device" memory is host memory, streams are synchronous) ----
extern "C" {
hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipDeviceGetPCIBusId(char *buf, int len, int dev) { snprintf(buf, (size_t)len, "0000:%02x:00.0", 5 + dev); return hipSuccess; }
hipError_t hipMalloc(void **p, size_t n) { *p = malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipFree(void *p) { free(p); return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = (hipStream_t)malloc(8); return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) { free(s); return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { *e = (hipEvent_t)malloc(8); return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { free(e); return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
const char *hipGetErrorString(hipError_t) { return "fake hip"; }
}
namespace crn {
static thread_local std::string g_err;
int fail(int code, const std::string &msg) { g_err = msg; return code; }
}  // namespace crn
extern "C" const char *crn_last_error(void) { return crn::g_err.c_str(); }

static int g_failures = 0;
#define REQUIRE(c)                                                                         \
  do {                                                                                     \
    if (!(c)) {                                                                            \
      fprintf(stderr, "comm_unit: line %d: %s FAILED (%s)\n", __LINE__, #c, crn_last_error()); \
      g_failures++;                                                                        \
      return;                                                                              \
    }                                                                                      \
  } while (0)

static void rank_main(int rank, int world, const uint8_t *id, int64_t bytes, int depth) {
  crn_comm *c = NULL;
  REQUIRE(crn_comm_create(0, rank, world, id, bytes, depth, &c) == CRN_OK);
  uint8_t *slot_ptr[16] = {NULL};
  for (int64_t step = 0; step < 3 * depth + 2; step++) {
    uint8_t *local = NULL;
    REQUIRE(crn_comm_local(c, step, NULL, &local) == CRN_OK);
    if (step < depth) slot_ptr[step] = local;
    REQUIRE(local == slot_ptr[step % depth]);                             // `depth` slots, taken in turn, reused
    for (int64_t other = 0; other < step && other < depth; other++)
      if (other != step % depth) REQUIRE(slot_ptr[other] != local);       // and distinct
    for (int64_t i = 0; i < bytes; i++) local[i] = (uint8_t)(100 * rank + 10 * step + (i & 7));
    REQUIRE(crn_comm_allgather(c, step, NULL) == CRN_OK);
    REQUIRE(crn_comm_finish(c, NULL) == CRN_OK);
    const uint8_t *all = NULL;
    REQUIRE(crn_comm_gathered(c, step, &all) == CRN_OK);
    for (int r = 0; r < world; r++)
      for (int64_t i = 0; i < bytes; i++) REQUIRE(all[(size_t)r * bytes + i] == (uint8_t)(100 * r + 10 * step + (i & 7)));   // rank order
  }
  // what the collective itself says it is (the stand-in counts the ranks that joined its group)
  crn_comm_info_t info;
  REQUIRE(crn_comm_info(c, &info) == CRN_OK);
  REQUIRE(info.nranks == world && info.rank == rank && info.depth == depth && info.bytes_per_rank == bytes);
  REQUIRE(info.gathers == 3 * depth + 2 && info.rccl_version == 0 && strstr(info.library, "fake_rccl") != NULL);
  REQUIRE(strncmp(info.pci_bus_id, "0000:", 5) == 0 && strlen(info.pci_bus_id) == 12);   // the device behind RCCL's (here: crn_comm_create's) ordinal
  REQUIRE(crn_comm_info(c, NULL) == CRN_ERR_ARG && crn_comm_info(NULL, &info) == CRN_ERR_ARG);
  uint8_t *p = NULL;
  REQUIRE(crn_comm_local(c, -1, NULL, &p) == CRN_ERR_ARG);
  REQUIRE(crn_comm_destroy(c) == CRN_OK);
}

int main(int argc, char **argv) {
  if (argc < 2) {
    fprintf(stderr, "usage: %s path/to/libfake_rccl.so\n", argv[0]);
    return 2;
  }
  setenv("CRN_RCCL_LIB", argv[1], 1);
  uint8_t id[CRN_COMM_ID_BYTES];
  if (crn_comm_unique_id(id) != CRN_OK) {
    fprintf(stderr, "comm_unit: %s\n", crn_last_error());
    return 1;
  }
  crn_comm *bad = NULL;
  if (crn_comm_create(0, 2, 2, id, 64, 2, &bad) != CRN_ERR_ARG || crn_comm_create(0, 0, 2, id, 0, 2, &bad) != CRN_ERR_ARG) return 1;
  const int worlds[] = {1, 2, 3, 5, 8}, depths[] = {1, 2, 3};
  for (int world : worlds)
    for (int depth : depths) {
      uint8_t gid[CRN_COMM_ID_BYTES];
      if (crn_comm_unique_id(gid) != CRN_OK) return 1;
      std::vector<std::thread> th;
      for (int r = 0; r < world; r++) th.emplace_back(rank_main, r, world, gid, (int64_t)(4 * 64 + 3 * depth), depth);   // ~4 epochs x 64 channels
      for (std::thread &t : th) t.join();
    }
  if (g_failures) return 1;
  printf("comm_unit: ok\n");
  return 0;
}
