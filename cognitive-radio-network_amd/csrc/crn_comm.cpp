// crn_comm.cpp — the one collective of the multi-GPU path (include/crn_sense.h, "multi-GPU"): an
// all-gather of each rank's per-epoch occupancy block over RCCL (xGMI inside a node), double-buffered
// and queued on a side stream so that it overlaps the next sensing launch.
//
// The path shards by stream with no data-path collective (no state crosses streams);
// every node's engine needs the whole occupancy picture to pick a free channel, hence this gather.
// Messages are a few KiB to ~100 KiB per rank: latency-bound, so one collective per batch, never per epoch.
//
// RCCL is bound at run time (dlopen) — a one-GPU CRTS node links libcrnsense without it — through the
// declarations of <rccl/rccl.h>.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/crn_sense.h"
#include "crn_internal.h"

static_assert(CRN_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "crn_sense.h carries the size of ncclUniqueId");

#define HIP_TRY(expr)                                                                      \
  do {                                                                                     \
    hipError_t _e = (expr);                                                                \
    if (_e != hipSuccess)                                                                  \
      return crn::fail(CRN_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(_e)); \
  } while (0)

namespace {

struct Rccl {
  void *lib = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  // what crn_comm_info reports (a library without them still gathers: the fields then read -1)
  decltype(&ncclCommCount) CommCount = nullptr;
  decltype(&ncclCommUserRank) CommUserRank = nullptr;
  decltype(&ncclCommCuDevice) CommCuDevice = nullptr;
  decltype(&ncclGetVersion) GetVersion = nullptr;
  std::string path;   // the name dlopen took
  std::string error;
};

Rccl &rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    // $CRN_RCCL_LIB, when set, is THE library: a name that does not load is an error, not a hint (a job told to use one RCCL must
    // not quietly run over another)
    const char *env = std::getenv("CRN_RCCL_LIB");
    const bool forced = env && *env;
    const char *names[] = {env, forced ? nullptr : "librccl.so.1", forced ? nullptr : "librccl.so", forced ? nullptr : "/opt/rocm/lib/librccl.so.1"};
    for (const char *n : names) {
      if (!n || !*n) continue;
      r.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
      if (r.lib) {
        r.path = n;
        break;
      }
      r.error = dlerror();
    }
    if (!r.lib) return;
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(r.lib, "ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(r.lib, "ncclCommInitRank"));
    r.AllGather = reinterpret_cast<decltype(r.AllGather)>(dlsym(r.lib, "ncclAllGather"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(r.lib, "ncclCommDestroy"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(r.lib, "ncclGetErrorString"));
    r.CommCount = reinterpret_cast<decltype(r.CommCount)>(dlsym(r.lib, "ncclCommCount"));
    r.CommUserRank = reinterpret_cast<decltype(r.CommUserRank)>(dlsym(r.lib, "ncclCommUserRank"));
    r.CommCuDevice = reinterpret_cast<decltype(r.CommCuDevice)>(dlsym(r.lib, "ncclCommCuDevice"));
    r.GetVersion = reinterpret_cast<decltype(r.GetVersion)>(dlsym(r.lib, "ncclGetVersion"));
    if (!r.GetUniqueId || !r.CommInitRank || !r.AllGather || !r.CommDestroy || !r.GetErrorString) {
      r.error = "librccl lacks ncclGetUniqueId / ncclCommInitRank / ncclAllGather / ncclCommDestroy";
      dlclose(r.lib);
      r.lib = nullptr;
    }
  });
  return r;
}

int need_rccl() {
  if (rccl().lib) return CRN_OK;
  return crn::fail(CRN_ERR_DEVICE, "RCCL is not available (" + rccl().error + "); set CRN_RCCL_LIB to librccl.so");
}

#define NCCL_TRY(expr)                                                                              \
  do {                                                                                              \
    ncclResult_t _r = (expr);                                                                       \
    if (_r != ncclSuccess)                                                                          \
      return crn::fail(CRN_ERR_DEVICE, std::string(#expr) + ": " + rccl().GetErrorString(_r));      \
  } while (0)

}  // namespace

struct crn_comm {
  int device = 0, rank = 0, world = 1, depth = 2;
  int64_t bytes = 0;             // per rank per slot
  ncclComm_t comm = nullptr;
  hipStream_t side = nullptr;    // the gathers run here, behind an event of the launch stream
  uint8_t *d_local = nullptr;    // [depth][bytes]
  uint8_t *d_all = nullptr;      // [depth][world][bytes]
  std::vector<hipEvent_t> ready; // launch stream: the slot's block has been written
  std::vector<hipEvent_t> done;  // side stream: the slot's gather has finished
  std::vector<char> pending;
  int64_t n_gathers = 0;         // all-gathers queued so far (crn_comm_info)
};

extern "C" {

int crn_comm_unique_id(uint8_t id[CRN_COMM_ID_BYTES]) {
  if (!id) return crn::fail(CRN_ERR_ARG, "null id");
  if (int rc = need_rccl()) return rc;
  ncclUniqueId u;
  NCCL_TRY(rccl().GetUniqueId(&u));
  std::memcpy(id, u.internal, CRN_COMM_ID_BYTES);
  return CRN_OK;
}

int crn_comm_create(int32_t device, int32_t rank, int32_t world, const uint8_t id[CRN_COMM_ID_BYTES],
                    int64_t bytes_per_rank, int32_t depth, crn_comm **out) {
  if (!out || !id) return crn::fail(CRN_ERR_ARG, "crn_comm_create: null argument");
  *out = nullptr;
  if (world < 1 || rank < 0 || rank >= world) return crn::fail(CRN_ERR_ARG, "rank / world out of range");
  if (bytes_per_rank < 1 || depth < 1 || depth > 16) return crn::fail(CRN_ERR_ARG, "bytes_per_rank < 1 or depth not in 1..16");
  if (int rc = need_rccl()) return rc;
  HIP_TRY(hipSetDevice(device));
  crn_comm *c = new (std::nothrow) crn_comm();
  if (!c) return crn::fail(CRN_ERR_NOMEM, "out of host memory");
  c->device = device;
  c->rank = rank;
  c->world = world;
  c->depth = depth;
  c->bytes = bytes_per_rank;
  c->pending.assign(depth, 0);
  hipError_t e = hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&c->d_local), (size_t)depth * bytes_per_rank);
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&c->d_all), (size_t)depth * world * bytes_per_rank);
  for (int i = 0; i < depth && e == hipSuccess; i++) {
    hipEvent_t a = nullptr, b = nullptr;
    e = hipEventCreateWithFlags(&a, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&b, hipEventDisableTiming);
    c->ready.push_back(a);
    c->done.push_back(b);
  }
  if (e != hipSuccess) {
    crn_comm_destroy(c);
    return crn::fail(CRN_ERR_NOMEM, std::string("crn_comm_create: ") + hipGetErrorString(e));
  }
  ncclUniqueId u;
  std::memcpy(u.internal, id, CRN_COMM_ID_BYTES);
  const ncclResult_t r = rccl().CommInitRank(&c->comm, world, u, rank);  // collective: every rank calls it
  if (r != ncclSuccess) {
    c->comm = nullptr;
    crn_comm_destroy(c);
    return crn::fail(CRN_ERR_DEVICE, std::string("ncclCommInitRank: ") + rccl().GetErrorString(r));
  }
  *out = c;
  return CRN_OK;
}

int crn_comm_info(crn_comm *c, crn_comm_info_t *out) {
  if (!c || !out) return crn::fail(CRN_ERR_ARG, "crn_comm_info: null argument");
  std::memset(out, 0, sizeof(*out));
  out->nranks = out->rank = out->rccl_device = out->rccl_version = -1;
  out->device = c->device;
  out->depth = c->depth;
  out->bytes_per_rank = c->bytes;
  out->gathers = c->n_gathers;
  Rccl &r = rccl();
  // asked of the communicator RCCL built, not echoed from crn_comm_create's arguments: what the collective itself believes
  if (r.CommCount) { int v = -1; NCCL_TRY(r.CommCount(c->comm, &v)); out->nranks = v; }
  if (r.CommUserRank) { int v = -1; NCCL_TRY(r.CommUserRank(c->comm, &v)); out->rank = v; }
  if (r.CommCuDevice) { int v = -1; NCCL_TRY(r.CommCuDevice(c->comm, &v)); out->rccl_device = v; }
  if (r.GetVersion) { int v = -1; NCCL_TRY(r.GetVersion(&v)); out->rccl_version = v; }
  std::snprintf(out->library, sizeof(out->library), "%s", r.path.c_str());
  // the physical device behind the ordinal (ranks isolated by HIP_VISIBLE_DEVICES all call theirs 0)
  if (hipDeviceGetPCIBusId(out->pci_bus_id, (int)sizeof(out->pci_bus_id), out->rccl_device >= 0 ? out->rccl_device : c->device) != hipSuccess)
    out->pci_bus_id[0] = 0;
  return CRN_OK;
}

int crn_comm_local(crn_comm *c, int64_t step, void *stream, uint8_t **d_local) {
  if (!c || !d_local || step < 0) return crn::fail(CRN_ERR_ARG, "crn_comm_local: bad argument");
  const int s = (int)(step % c->depth);
  if (c->pending[s]) {  // the slot's previous gather still reads it: the launch stream waits on the device, not the host
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamWaitEvent(static_cast<hipStream_t>(stream), c->done[s], 0));
    c->pending[s] = 0;
  }
  *d_local = c->d_local + (size_t)s * c->bytes;
  return CRN_OK;
}

int crn_comm_local_addr(crn_comm *c, int64_t step, uint8_t **d_local) {
  if (!c || !d_local || step < 0) return crn::fail(CRN_ERR_ARG, "crn_comm_local_addr: bad argument");
  *d_local = c->d_local + (size_t)(step % c->depth) * c->bytes;
  return CRN_OK;
}

int crn_comm_wait(crn_comm *c, int64_t step, void *stream) {
  if (!c || step < 0) return crn::fail(CRN_ERR_ARG, "crn_comm_wait: bad argument");
  const int s = (int)(step % c->depth);
  if (!c->pending[s]) return CRN_OK;     // no gather queued on this slot since it was last handed out
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipStreamWaitEvent(static_cast<hipStream_t>(stream), c->done[s], 0));
  return CRN_OK;                          // the slot stays pending: crn_comm_local / crn_comm_finish still order their stream behind it
}

int crn_comm_allgather(crn_comm *c, int64_t step, void *stream) {
  if (!c || step < 0) return crn::fail(CRN_ERR_ARG, "crn_comm_allgather: bad argument");
  const int s = (int)(step % c->depth);
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipEventRecord(c->ready[s], static_cast<hipStream_t>(stream)));
  HIP_TRY(hipStreamWaitEvent(c->side, c->ready[s], 0));
  NCCL_TRY(rccl().AllGather(c->d_local + (size_t)s * c->bytes, c->d_all + (size_t)s * c->world * c->bytes,
                            (size_t)c->bytes, ncclUint8, c->comm, c->side));
  HIP_TRY(hipEventRecord(c->done[s], c->side));
  c->pending[s] = 1;
  c->n_gathers++;
  return CRN_OK;
}

int crn_comm_gathered(crn_comm *c, int64_t step, const uint8_t **d_all) {
  if (!c || !d_all || step < 0) return crn::fail(CRN_ERR_ARG, "crn_comm_gathered: bad argument");
  *d_all = c->d_all + (size_t)(step % c->depth) * c->world * c->bytes;
  return CRN_OK;
}

int crn_comm_finish(crn_comm *c, void *stream) {
  if (!c) return crn::fail(CRN_ERR_ARG, "null communicator");
  HIP_TRY(hipSetDevice(c->device));
  for (int s = 0; s < c->depth; s++)
    if (c->pending[s]) {
      HIP_TRY(hipStreamWaitEvent(static_cast<hipStream_t>(stream), c->done[s], 0));
      c->pending[s] = 0;
    }
  return CRN_OK;
}

int crn_comm_destroy(crn_comm *c) {
  if (!c) return CRN_OK;
  (void)hipSetDevice(c->device);
  if (c->side) (void)hipStreamSynchronize(c->side);
  if (c->comm) (void)rccl().CommDestroy(c->comm);
  for (hipEvent_t e : c->ready)
    if (e) (void)hipEventDestroy(e);
  for (hipEvent_t e : c->done)
    if (e) (void)hipEventDestroy(e);
  if (c->d_local) (void)hipFree(c->d_local);
  if (c->d_all) (void)hipFree(c->d_all);
  if (c->side) (void)hipStreamDestroy(c->side);
  delete c;
  return CRN_OK;
}

}  // extern "C"
