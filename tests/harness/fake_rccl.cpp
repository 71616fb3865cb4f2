// fake_rccl.cpp -> libfake_rccl.so — TEST INFRASTRUCTURE.  The RCCL entry points csrc/crn_comm.cpp binds at run time,
// for "ranks" that are THREADS of one CPU process (tests/harness/comm_unit.cpp, loaded through $CRN_RCCL_LIB): the all-gather
// really places rank r's block at offset r * count of every rank's receive buffer, so the test sees what a world of two ranks
// sees — rank order, slot addresses, byte counts — without a GPU.  Streams are ignored (the stand-in HIP calls are synchronous).
#include <condition_variable>
#include <cstring>
#include <map>
#include <mutex>
#include <random>
#include <string>
#include <vector>

typedef enum { ncclSuccess = 0, ncclInvalidArgument = 4 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclChar = 0, ncclUint8 = 1 } ncclDataType_t;
typedef struct { char internal[128]; } ncclUniqueId;

namespace {
struct Group {
  int nranks = 0, joined = 0, arrived = 0, left = 0;
  long long round = 0;
  std::vector<const void *> send;
  std::mutex mu;
  std::condition_variable cv;
};
struct Comm { Group *g; int rank; };
std::mutex g_mu;
std::map<std::string, Group *> g_groups;

void barrier(Group *g, std::unique_lock<std::mutex> &lk) {
  const long long my = g->round;
  if (++g->arrived == g->nranks) {
    g->arrived = 0;
    g->round++;
    g->cv.notify_all();
  } else {
    g->cv.wait(lk, [&] { return g->round != my; });
  }
}
}  // namespace

extern "C" {
#define VIS __attribute__((visibility("default")))
VIS ncclResult_t ncclGetUniqueId(ncclUniqueId *id) {
  static std::mt19937_64 rng(12345);
  std::lock_guard<std::mutex> lk(g_mu);
  for (int i = 0; i < 128; i += 8) {
    const unsigned long long v = rng();
    memcpy(id->internal + i, &v, 8);
  }
  return ncclSuccess;
}
VIS ncclResult_t ncclCommInitRank(Comm **comm, int nranks, ncclUniqueId id, int rank) {
  if (rank < 0 || rank >= nranks) return ncclInvalidArgument;
  Group *g;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    Group *&slot = g_groups[std::string(id.internal, 128)];
    if (!slot) {
      slot = new Group();
      slot->nranks = nranks;
      slot->send.assign(nranks, nullptr);
    }
    g = slot;
  }
  std::unique_lock<std::mutex> lk(g->mu);
  if (g->nranks != nranks) return ncclInvalidArgument;
  g->joined++;
  g->cv.notify_all();
  g->cv.wait(lk, [&] { return g->joined >= g->nranks; });   // collective: returns once every rank has called it
  *comm = new Comm{g, rank};
  return ncclSuccess;
}
VIS ncclResult_t ncclAllGather(const void *sendbuff, void *recvbuff, size_t count, ncclDataType_t type, Comm *c, void *) {
  if (type != ncclUint8 && type != ncclInt8) return ncclInvalidArgument;
  Group *g = c->g;
  std::unique_lock<std::mutex> lk(g->mu);
  g->send[c->rank] = sendbuff;
  barrier(g, lk);                                            // every rank's send pointer is published
  for (int r = 0; r < g->nranks; r++) memcpy(static_cast<char *>(recvbuff) + (size_t)r * count, g->send[r], count);
  barrier(g, lk);                                            // nobody reuses its send buffer before all have copied
  return ncclSuccess;
}
VIS ncclResult_t ncclCommDestroy(Comm *c) { delete c; return ncclSuccess; }
VIS const char *ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : "fake rccl: invalid argument"; }
// the queries behind crn_comm_info: the count is the number of ranks that really joined the group
VIS ncclResult_t ncclCommCount(Comm *c, int *n) { std::lock_guard<std::mutex> lk(c->g->mu); *n = c->g->joined; return ncclSuccess; }
VIS ncclResult_t ncclCommUserRank(Comm *c, int *r) { *r = c->rank; return ncclSuccess; }
VIS ncclResult_t ncclCommCuDevice(Comm *, int *d) { *d = 0; return ncclSuccess; }
VIS ncclResult_t ncclGetVersion(int *v) { *v = 0; return ncclSuccess; }   // 0: a stand-in, no RCCL release
}
