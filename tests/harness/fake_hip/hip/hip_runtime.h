// TEST INFRASTRUCTURE: a host-only stand-in for the handful of HIP runtime calls csrc/crn_ingest.cpp makes, so that
// the ring's HOST logic (slot hand-out, single copy, hand-off to the launcher thread, BUSY refusals, flush / drain,
// result order) can be unit-tested — under ThreadSanitizer — on a machine without a GPU (tests/harness/ring_unit.cpp).
// "Device" memory is host memory, copies are memcpy, an event becomes ready g_fake_gpu_latency_ns after it is recorded.
// Nothing of this is ever linked into the product.
#ifndef CRN_FAKE_HIP_RUNTIME_H
#define CRN_FAKE_HIP_RUNTIME_H
#include <atomic>
#include <chrono>
#include <cstddef>
#include <cstdlib>
#include <cstring>
#include <thread>

typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorNotReady = 600, hipErrorNotSupported = 801 };
enum { hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2 };
enum { hipHostMallocDefault = 0, hipStreamNonBlocking = 1, hipEventDisableTiming = 2 };
struct fake_hip_stream { int unused; };
struct fake_hip_event { std::atomic<long long> ready_at_ns; };
typedef fake_hip_stream *hipStream_t;
typedef fake_hip_event *hipEvent_t;

extern std::atomic<long long> g_fake_gpu_latency_ns;   // defined by the test: how long a "batch" stays on the "GPU"

// A test marks its own thread (fake_hip_watch_thread = true) around the calls it wants to prove HIP-free — the engine's execute()
// runs with CE_mutex held and must only enqueue: every stand-in below counts a call made from a marked thread, and the two that
// can block count separately.  (thread_local inline variables: one instance per thread across translation units, C++17.)
inline thread_local bool fake_hip_watch_thread = false;
inline std::atomic<long long> fake_hip_calls_on_watched_threads{0}, fake_hip_waits_on_watched_threads{0};
inline void fake_hip_note(bool blocking = false) {
  if (fake_hip_watch_thread) {
    fake_hip_calls_on_watched_threads++;
    if (blocking) fake_hip_waits_on_watched_threads++;
  }
}

inline long long fake_hip_now_ns() {
  return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
inline const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "success" : e == hipErrorNotReady ? "not ready" : "fake hip error"; }
inline hipError_t hipSetDevice(int) { fake_hip_note(); return hipSuccess; }
inline hipError_t hipMalloc(void **p, size_t n) { fake_hip_note(); *p = std::malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
inline hipError_t hipHostMalloc(void **p, size_t n, unsigned) { fake_hip_note(); return hipMalloc(p, n); }
inline hipError_t hipFree(void *p) { fake_hip_note(); std::free(p); return hipSuccess; }
inline hipError_t hipHostFree(void *p) { fake_hip_note(); std::free(p); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, int, hipStream_t) { fake_hip_note(); std::memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { fake_hip_note(); *s = new fake_hip_stream(); return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t s) { fake_hip_note(); delete s; return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { fake_hip_note(true); return hipSuccess; }
inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { fake_hip_note(); *e = new fake_hip_event(); (*e)->ready_at_ns.store(0); return hipSuccess; }
inline hipError_t hipEventDestroy(hipEvent_t e) { fake_hip_note(); delete e; return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { fake_hip_note(); e->ready_at_ns.store(fake_hip_now_ns() + g_fake_gpu_latency_ns.load()); return hipSuccess; }
inline hipError_t hipEventQuery(hipEvent_t e) { fake_hip_note(); return fake_hip_now_ns() >= e->ready_at_ns.load() ? hipSuccess : hipErrorNotReady; }
inline void (*g_fake_hip_on_event_sync)() = nullptr;   // a test's hook: runs inside the wait, i.e. while the caller holds no lock of its own
inline hipError_t hipEventSynchronize(hipEvent_t e) {
  fake_hip_note(true);
  if (g_fake_hip_on_event_sync) g_fake_hip_on_event_sync();
  const bool w = fake_hip_watch_thread;
  fake_hip_watch_thread = false;   // (the polling below is this wait, not further calls)
  struct Restore { bool w; ~Restore() { fake_hip_watch_thread = w; } } restore{w};
  while (hipEventQuery(e) != hipSuccess) std::this_thread::sleep_for(std::chrono::microseconds(20));
  return hipSuccess;
}

// What csrc/crn_api.cpp needs beyond the ring's calls (tests/harness/api_unit.cpp: table building and launch geometry on the host).
// The "device" has g_fake_hip_cus compute units; kernels are the test's own stand-ins for the launch_* functions.
#define HIP_VERSION_MAJOR 7
#define HIP_VERSION_MINOR 2
#define HIP_VERSION_PATCH 0
#define HIP_VERSION (HIP_VERSION_MAJOR * 10000000 + HIP_VERSION_MINOR * 100000 + HIP_VERSION_PATCH)
struct float2 { float x, y; };
inline float2 make_float2(float x, float y) { return float2{x, y}; }
template <class T> inline hipError_t hipMalloc(T **p, size_t n) { return hipMalloc(reinterpret_cast<void **>(p), n); }
enum { hipDeviceAttributeMultiprocessorCount = 63, hipDeviceAttributeMaxSharedMemoryPerBlock = 74 };
inline int g_fake_hip_cus = 256;
inline int g_fake_hip_lds_bytes = 160 * 1024;   // gfx950; a test may shrink it (a device or partition with 64 KiB)
inline hipError_t hipGetDeviceCount(int *n) { *n = 1; return hipSuccess; }
inline hipError_t hipDeviceGetAttribute(int *v, int attr, int) {
  *v = attr == hipDeviceAttributeMultiprocessorCount ? g_fake_hip_cus : attr == hipDeviceAttributeMaxSharedMemoryPerBlock ? g_fake_hip_lds_bytes : 0;
  return hipSuccess;
}
// stream capture: a test marks a stream as capturing (crn_api.cpp refuses updates on it)
enum hipStreamCaptureStatus { hipStreamCaptureStatusNone = 0, hipStreamCaptureStatusActive = 1 };
inline std::atomic<void *> g_fake_hip_capturing_stream{nullptr};
inline hipError_t hipStreamIsCapturing(hipStream_t s, hipStreamCaptureStatus *st) {
  *st = (s != nullptr && (void *)s == g_fake_hip_capturing_stream.load()) ? hipStreamCaptureStatusActive : hipStreamCaptureStatusNone;
  return hipSuccess;
}
inline hipError_t hipRuntimeGetVersion(int *v) { *v = HIP_VERSION; return hipSuccess; }
inline hipError_t hipMemcpy(void *d, const void *s, size_t n, int) { fake_hip_note(true); std::memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipEventCreate(hipEvent_t *e) { return hipEventCreateWithFlags(e, 0); }
inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b) { *ms = (float)((b->ready_at_ns.load() - a->ready_at_ns.load()) * 1e-6); return hipSuccess; }
#endif
