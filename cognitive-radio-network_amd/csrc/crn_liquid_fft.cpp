// crn_liquid_fft.cpp — libcrnliquidfft.so: liquid-dsp's fft_create_plan / fft_execute /
// fft_destroy_plan (include/crn_liquid_fft.h) over crn_fft_forward_device.  Host C++, links
// libcrnsense.so through its C ABI and the HIP runtime for the two copies.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#include "../../include/crn_liquid_fft.h"
#include "../../include/crn_sense.h"

struct fftplan_s {
  unsigned n;
  liquid_float_complex *x, *y;
  crn_handle *h;
  float *d_in, *d_out;
  hipStream_t stream;
  bool pinned_x, pinned_y;  // the caller's arrays page-locked for the copies (best effort)
};

namespace {
[[noreturn]] void die(const char *what, const char *detail) {
  // the liquid API has no error return; CRTS's convention for set-up failures (src/crts.cpp:111-115)
  std::fprintf(stderr, "crnliquidfft: %s%s%s\n", what, detail ? ": " : "", detail ? detail : "");
  std::exit(EXIT_FAILURE);
}
}  // namespace

extern "C" fftplan fft_create_plan(unsigned int n, liquid_float_complex *x, liquid_float_complex *y, int dir, int /*flags*/) {
  if (dir != LIQUID_FFT_FORWARD) die("only LIQUID_FFT_FORWARD is provided (the sensing path's direction)", nullptr);
  if (n != 512 && n != 1024 && n != 2048 && n != 4096) die("fft_create_plan: n must be 512, 1024, 2048 or 4096", nullptr);
  if (!x || !y) die("fft_create_plan: null buffer", nullptr);
  fftplan p = new fftplan_s();
  p->n = n;
  p->x = x;
  p->y = y;
  crn_cfg cfg;
  if (crn_cfg_energy_scaled(&cfg, (int32_t)n, 4.0f) != CRN_OK) die("crn_cfg_energy_scaled", crn_last_error());
  const char *dev = std::getenv("CRN_DEVICE");
  cfg.device = dev ? std::atoi(dev) : 0;
  if (crn_sense_create(&cfg, &p->h) != CRN_OK) die("crn_sense_create", crn_last_error());
  if (hipSetDevice(cfg.device) != hipSuccess || hipMalloc(&p->d_in, sizeof(float) * 2 * n) != hipSuccess ||
      hipMalloc(&p->d_out, sizeof(float) * 2 * n) != hipSuccess || hipStreamCreate(&p->stream) != hipSuccess)
    die("device buffers", hipGetErrorString(hipGetLastError()));
  // page-lock the bound arrays: the two copies of every fft_execute then go straight over DMA
  p->pinned_x = hipHostRegister(x, sizeof(float) * 2 * n, hipHostRegisterDefault) == hipSuccess;
  p->pinned_y = hipHostRegister(y, sizeof(float) * 2 * n, hipHostRegisterDefault) == hipSuccess;
  (void)hipGetLastError();
  return p;
}

extern "C" void fft_execute(fftplan p) {
  if (!p) die("fft_execute: null plan", nullptr);
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
  if (p->pinned_x) (void)hipHostUnregister(p->x);
  if (p->pinned_y) (void)hipHostUnregister(p->y);
  (void)hipStreamDestroy(p->stream);
  (void)hipFree(p->d_in);
  (void)hipFree(p->d_out);
  (void)crn_sense_destroy(p->h);
  delete p;
}
