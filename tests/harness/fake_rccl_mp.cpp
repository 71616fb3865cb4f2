// fake_rccl_mp.cpp -> libfake_rccl_mp.so — TEST INFRASTRUCTURE.  The RCCL entry points csrc/crn_comm.cpp binds at run time, for
// ranks that are PROCESSES sharing one GPU box ($CRN_RCCL_LIB; tests/test_bench_cli.py runs `torch.distributed.run --nproc-per-node 2
// bench.py --gpus 2` with it): real RCCL refuses two ranks on one device, and the pool hands out one GPU, so this is how bench.py's
// whole N > 1 flow (gloo control plane, unique-id broadcast, sharding, barriers, max-over-ranks timing, the gathered-vector check)
// runs on hardware before the driver's 8-GPU run.  The all-gather goes through a POSIX shared-memory segment named after the unique
// id: every rank copies its block down (after draining the stream it was queued behind), a barrier, every rank copies all blocks
// up, a barrier.  Blocking and slow on purpose; only the data placement and ordering semantics are RCCL's.
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <sys/mman.h>
#include <unistd.h>

typedef enum { ncclSuccess = 0, ncclSystemError = 2, ncclInvalidArgument = 4 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclChar = 0, ncclUint8 = 1 } ncclDataType_t;
typedef struct { char internal[128]; } ncclUniqueId;

namespace {
constexpr size_t kMaxRanks = 16, kMaxBytesPerRank = 8u << 20;
struct Shared {
  volatile int count, generation;   // sense-reversing barrier
  volatile int joined;
  char data[kMaxRanks * kMaxBytesPerRank];
};
struct Comm {
  Shared *sh;
  int rank, nranks;
  char name[64];
};
void barrier(Comm *c) {
  const int gen = c->sh->generation;
  if (__atomic_add_fetch(&c->sh->count, 1, __ATOMIC_ACQ_REL) == c->nranks) {
    c->sh->count = 0;
    __atomic_add_fetch(&c->sh->generation, 1, __ATOMIC_ACQ_REL);
  } else {
    while (__atomic_load_n(&c->sh->generation, __ATOMIC_ACQUIRE) == gen) usleep(20);
  }
}
}  // namespace

extern "C" {
#define VIS __attribute__((visibility("default")))
VIS ncclResult_t ncclGetUniqueId(ncclUniqueId *id) {
  memset(id, 0, sizeof(*id));
  FILE *f = fopen("/dev/urandom", "rb");
  if (!f || fread(id->internal, 1, 16, f) != 16) return ncclSystemError;
  fclose(f);
  return ncclSuccess;
}
VIS ncclResult_t ncclCommInitRank(Comm **out, int nranks, ncclUniqueId id, int rank) {
  if (rank < 0 || rank >= nranks || nranks > (int)kMaxRanks) return ncclInvalidArgument;
  Comm *c = new Comm();
  c->rank = rank;
  c->nranks = nranks;
  char hex[33];
  for (int i = 0; i < 16; i++) snprintf(hex + 2 * i, 3, "%02x", (unsigned char)id.internal[i]);
  snprintf(c->name, sizeof(c->name), "/crn_fake_rccl_%s", hex);
  const int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
  if (fd < 0 || ftruncate(fd, sizeof(Shared)) != 0) return ncclSystemError;   // a fresh segment is zero-filled
  c->sh = static_cast<Shared *>(mmap(NULL, sizeof(Shared), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0));
  close(fd);
  if (c->sh == MAP_FAILED) return ncclSystemError;
  __atomic_add_fetch(&c->sh->joined, 1, __ATOMIC_ACQ_REL);
  while (__atomic_load_n(&c->sh->joined, __ATOMIC_ACQUIRE) < nranks) usleep(100);   // collective, like the real one
  *out = c;
  return ncclSuccess;
}
VIS ncclResult_t ncclAllGather(const void *send, void *recv, size_t count, ncclDataType_t type, Comm *c, hipStream_t stream) {
  if ((type != ncclUint8 && type != ncclInt8) || count > kMaxBytesPerRank) return ncclInvalidArgument;
  if (hipStreamSynchronize(stream) != hipSuccess) return ncclSystemError;    // everything queued before the gather has happened
  if (hipMemcpy(c->sh->data + (size_t)c->rank * count, send, count, hipMemcpyDeviceToHost) != hipSuccess) return ncclSystemError;
  barrier(c);
  if (hipMemcpy(recv, c->sh->data, (size_t)c->nranks * count, hipMemcpyHostToDevice) != hipSuccess) return ncclSystemError;
  barrier(c);
  return ncclSuccess;
}
VIS ncclResult_t ncclCommDestroy(Comm *c) {
  if (c->rank == 0) shm_unlink(c->name);
  munmap(c->sh, sizeof(Shared));
  delete c;
  return ncclSuccess;
}
VIS const char *ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : "fake rccl (multi-process): error"; }
// the queries behind crn_comm_info: the count is the number of processes that really attached to the segment, not nranks echoed
VIS ncclResult_t ncclCommCount(Comm *c, int *n) { *n = __atomic_load_n(&c->sh->joined, __ATOMIC_ACQUIRE); return ncclSuccess; }
VIS ncclResult_t ncclCommUserRank(Comm *c, int *r) { *r = c->rank; return ncclSuccess; }
VIS ncclResult_t ncclCommCuDevice(Comm *, int *d) { return hipGetDevice(d) == hipSuccess ? ncclSuccess : ncclSystemError; }
VIS ncclResult_t ncclGetVersion(int *v) { *v = 0; return ncclSuccess; }   // 0: a stand-in, no RCCL release
}
