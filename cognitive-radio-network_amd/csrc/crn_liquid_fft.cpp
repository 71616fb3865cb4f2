// crn_liquid_fft.cpp — libcrnliquidfft.so: liquid-dsp's fft_create_plan / fft_execute /
// fft_destroy_plan (include/crn_liquid_fft.h) over crn_fft_forward_device.  Host C++, links
// libcrnsense.so through its C ABI and the HIP runtime for the two copies.
//
// These are global symbols named like liquid's, so when the library precedes -lliquid on a link line
// EVERY call to them binds here — including liquid's own internal ones: the ECR constructor's
// ofdmflexframegen_create / ofdmflexframesync_create (reference: src/extensible_cognitive_radio.cpp:113,123)
// build M-subcarrier BACKWARD and forward plans through fft_create_plan.  Only the sensing path's plans
// are taken here (forward, N in {512, 1024, 2048, 4096}); every other plan is handed to the next
// definition in the search order (dlsym(RTLD_NEXT): liquid's own), and fft_execute / fft_destroy_plan
// dispatch on who created the plan.
#ifndef _GNU_SOURCE
#define _GNU_SOURCE
#endif
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <mutex>
#include <unordered_set>

#include "../../include/crn_liquid_fft.h"
#include "../../include/crn_sense.h"

struct fftplan_s {
  unsigned n;
  liquid_float_complex *x, *y;
  crn_handle *h;
  float *d_in, *d_out;
  hipStream_t stream;
  bool pinned_x, pinned_y;  // the caller's arrays page-locked for the copies (best effort)
  float *map_x, *map_y;     // both non-null: the device's view of the page-locked arrays — the kernel reads x and writes y itself
};

namespace {

[[noreturn]] void die(const char *what, const char *detail) {
  // the liquid API has no error return; CRTS's convention for set-up failures (src/crts.cpp:111-115)
  std::fprintf(stderr, "crnliquidfft: %s%s%s\n", what, detail ? ": " : "", detail ? detail : "");
  std::exit(EXIT_FAILURE);
}

// plans created here (anything else belongs to the next library and is a different struct)
std::mutex g_mu;
std::unordered_set<const void *> g_ours;
long g_forwarded = 0;  // plans handed to the next library

bool is_ours(const void *p) {
  std::lock_guard<std::mutex> lk(g_mu);
  return g_ours.count(p) != 0;
}

typedef fftplan (*create_fn)(unsigned int, liquid_float_complex *, liquid_float_complex *, int, int);
typedef void (*plan_fn)(fftplan);

template <class F>
F next_symbol(const char *name) {
  return reinterpret_cast<F>(dlsym(RTLD_NEXT, name));
}

bool sensing_plan(unsigned n, int dir) {
  return dir == LIQUID_FFT_FORWARD && (n == 512 || n == 1024 || n == 2048 || n == 4096);
}

}  // namespace

extern "C" fftplan fft_create_plan(unsigned int n, liquid_float_complex *x, liquid_float_complex *y, int dir, int flags) {
  if (!sensing_plan(n, dir)) {
    // not the sensing path's transform: liquid's own (its OFDM framing plans, any other user)
    static const create_fn next = next_symbol<create_fn>("fft_create_plan");
    if (next) {
      {
        std::lock_guard<std::mutex> lk(g_mu);
        g_forwarded++;
      }
      return next(n, x, y, dir, flags);
    }
    die("fft_create_plan: only forward transforms of 512, 1024, 2048 or 4096 points are provided and no other "
        "fft_create_plan follows libcrnliquidfft in the link order (name it before -lliquid)", nullptr);
  }
  if (!x || !y) die("fft_create_plan: null buffer", nullptr);
  fftplan p = new fftplan_s();
  p->n = n;
  p->x = x;
  p->y = y;
  crn_cfg cfg;
  if (crn_cfg_energy_scaled(&cfg, (int32_t)n, 4.0f) != CRN_OK) die("crn_cfg_energy_scaled", crn_last_error());
  const char *dev = std::getenv("CRN_DEVICE");
  cfg.device = dev ? std::atoi(dev) : 0;
  if (crn_sense_create(&cfg, &p->h) != CRN_OK) die("crn_sense_create", crn_last_error());  // no GPU: no CPU fallback
  if (hipSetDevice(cfg.device) != hipSuccess || hipMalloc(&p->d_in, sizeof(float) * 2 * n) != hipSuccess ||
      hipMalloc(&p->d_out, sizeof(float) * 2 * n) != hipSuccess || hipStreamCreate(&p->stream) != hipSuccess)
    die("device buffers", hipGetErrorString(hipGetLastError()));
  // page-lock the bound arrays: the two copies of every fft_execute then go straight over DMA
  p->pinned_x = hipHostRegister(x, sizeof(float) * 2 * n, hipHostRegisterMapped) == hipSuccess;
  p->pinned_y = hipHostRegister(y, sizeof(float) * 2 * n, hipHostRegisterMapped) == hipSuccess;
  // A frame is 4-32 KiB: instead of upload + launch + download, the kernel reads the caller's x and writes the caller's y over
  // the bus itself — one launch and one wait per fft_execute ($CRN_LIQUID_ZEROCOPY=0: the three-step form)
  p->map_x = p->map_y = nullptr;
  const char *zc = std::getenv("CRN_LIQUID_ZEROCOPY");
  if (p->pinned_x && p->pinned_y && !(zc && zc[0] == '0')) {
    void *dx = nullptr, *dy = nullptr;
    if (hipHostGetDevicePointer(&dx, x, 0) == hipSuccess && hipHostGetDevicePointer(&dy, y, 0) == hipSuccess && dx && dy &&
        (reinterpret_cast<uintptr_t>(dx) & 7u) == 0) {
      p->map_x = static_cast<float *>(dx);
      p->map_y = static_cast<float *>(dy);
    }
  }
  (void)hipGetLastError();
  {
    std::lock_guard<std::mutex> lk(g_mu);
    g_ours.insert(p);
  }
  return p;
}

extern "C" __attribute__((visibility("default"))) long crn_liquid_fft_forwarded(void) {
  std::lock_guard<std::mutex> lk(g_mu);
  return g_forwarded;
}

extern "C" void fft_execute(fftplan p) {
  if (!p) die("fft_execute: null plan", nullptr);
  if (!is_ours(p)) {
    static const plan_fn next = next_symbol<plan_fn>("fft_execute");
    if (!next) die("fft_execute: plan was not created by libcrnliquidfft and no other fft_execute follows it", nullptr);
    next(p);
    return;
  }
  if (p->map_x) {
    if (crn_fft_forward_device(p->h, p->map_x, 1, (int32_t)p->n, 0, p->map_y, p->stream) != CRN_OK)
      die("crn_fft_forward_device", crn_last_error());
    const hipError_t e = hipStreamSynchronize(p->stream);
    if (e != hipSuccess) die("fft_execute", hipGetErrorString(e));
    return;
  }
  const size_t bytes = sizeof(float) * 2 * p->n;
  hipError_t e = hipMemcpyAsync(p->d_in, p->x, bytes, hipMemcpyHostToDevice, p->stream);
  if (e == hipSuccess && crn_fft_forward_device(p->h, p->d_in, 1, (int32_t)p->n, 0, p->d_out, p->stream) != CRN_OK)
    die("crn_fft_forward_device", crn_last_error());
  if (e == hipSuccess) e = hipMemcpyAsync(p->y, p->d_out, bytes, hipMemcpyDeviceToHost, p->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(p->stream);
  if (e != hipSuccess) die("fft_execute", hipGetErrorString(e));
}

extern "C" void fft_destroy_plan(fftplan p) {
  if (!p) return;
  if (!is_ours(p)) {
    static const plan_fn next = next_symbol<plan_fn>("fft_destroy_plan");
    if (next) next(p);
    return;
  }
  {
    std::lock_guard<std::mutex> lk(g_mu);
    g_ours.erase(p);
  }
  if (p->pinned_x) (void)hipHostUnregister(p->x);
  if (p->pinned_y) (void)hipHostUnregister(p->y);
  (void)hipStreamDestroy(p->stream);
  (void)hipFree(p->d_in);
  (void)hipFree(p->d_out);
  (void)crn_sense_destroy(p->h);
  delete p;
}
