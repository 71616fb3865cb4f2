// crn_monitor.hip — the display side of the reference's GNU Radio monitor on the device
// (reference: spectrum_analyzer.py:262-275: qtgui.freq_sink_c(fft_size = 1024, firdes.WIN_BLACKMAN_hARRIS, ..),
// set_fft_average(0.1), plus the waterfall sink).  The windowed FFT + |X|^2 (+ K-frame mean) is the sensing
// kernel's `spectrum` output; this epilogue turns those rows into what the sinks draw: fftshifted dB rows (the
// waterfall) and the single-pole-IIR averaged trace, with the IIR state kept on the device between calls.
//
// GNU Radio itself is a third-party dependency absent from the reference tree; its published algorithm
// (gr-qtgui 3.7, freq_sink_c_impl.cc: window multiply -> FFT -> volk power_spectral_density
// 10 log10(|X / N|^2) -> fftshift -> d_magbuf = (1 - a) d_magbuf + a new, on the dB values) is what
// CRN_MONITOR_GNURADIO restates; CRN_MONITOR_PSD averages linear power normalised by N sum(w^2) instead.
#include <hip/hip_runtime.h>

#include "crn_kernels.h"

namespace crn {

// One thread per displayed column j (bin (j + N/2) mod N): rows are walked in order because the IIR is.
__global__ __launch_bounds__(256) void monitor_rows_kernel(const MonitorParams p) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= p.n) return;
  const int src = (j + p.n / 2) & (p.n - 1);   // fftshift: the centre frequency in the middle, like the sinks
  const float floor_p = 1e-30f;
  float acc = p.first ? 0.f : p.state[j];
  for (long long r = 0; r < p.n_rows; r++) {
    const float pw = p.spectrum[r * p.n + src] * p.scale;
    const float db = 10.0f * log10f(fmaxf(pw, floor_p));
    if (p.waterfall_db != nullptr) p.waterfall_db[r * p.n + j] = db;
    const float v = p.db_domain ? db : pw;
    acc = (p.first && r == 0) ? v : fmaf(p.alpha, v - acc, acc);   // (1 - a) acc + a v
    if (p.average_db != nullptr) p.average_db[r * p.n + j] = p.db_domain ? acc : 10.0f * log10f(fmaxf(acc, floor_p));
  }
  p.state[j] = acc;
}

hipError_t launch_monitor(const MonitorParams &p, hipStream_t stream) {
  if (p.n_rows <= 0) return hipSuccess;
  hipLaunchKernelGGL(monitor_rows_kernel, dim3((unsigned)((p.n + 255) / 256)), dim3(256), 0, stream, p);
  return hipGetLastError();
}

// ---- noise-floor estimate for threshold plans (crn_noise_floor_device, include/crn_sense.h) -------------------------------------
// thr_b = lambda x NF_est, NF_est = the median band energy.  Lower median (element (n - 1) / 2 of the sorted
// values) by rank counting: no sort, no scratch beyond one float per epoch, ties broken by index so that exactly one element wins.

// one thread per epoch: the median of its n_bands band energies
__global__ __launch_bounds__(256) void noise_floor_rows_kernel(const float *feat, int n_epochs, int nb, float *med) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_epochs) return;
  const float *v = feat + (long long)e * nb;
  const int want = (nb - 1) / 2;
  float m = __builtin_nanf("");   // a row holding a NaN has no median: the estimate says so instead of keeping a stale value
  for (int i = 0; i < nb; i++) {
    const float vi = v[i];
    int below = 0;
    for (int j = 0; j < nb; j++) below += (v[j] < vi) || (v[j] == vi && j < i);
    if (below == want) m = vi;
  }
  for (int j = 0; j < nb; j++)
    if (v[j] != v[j]) m = v[j];
  med[e] = m;
}

// one workgroup: the median of the per-epoch medians (n <= 4096)
__global__ __launch_bounds__(256) void noise_floor_final_kernel(const float *med, int n, float *out) {
  extern __shared__ float sm[];
  for (int i = threadIdx.x; i < n; i += blockDim.x) sm[i] = med[i];
  if (threadIdx.x == 0) *out = __builtin_nanf("");
  __syncthreads();
  const int want = (n - 1) / 2;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const float vi = sm[i];
    int below = 0;
    for (int j = 0; j < n; j++) below += (sm[j] < vi) || (sm[j] == vi && j < i);
    if (below == want) *out = vi;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += blockDim.x)
    if (sm[i] != sm[i]) *out = sm[i];   // any epoch without a median: neither has the batch
}

hipError_t launch_noise_floor(const float *feat, int n_epochs, int nb, float *scratch, hipStream_t stream) {
  hipLaunchKernelGGL(noise_floor_rows_kernel, dim3((unsigned)((n_epochs + 255) / 256)), dim3(256), 0, stream, feat, n_epochs, nb, scratch);
  hipLaunchKernelGGL(noise_floor_final_kernel, dim3(1), dim3(256), (size_t)n_epochs * sizeof(float), stream, scratch, n_epochs,
                     scratch + kNoiseFloorMaxEpochs);
  return hipGetLastError();
}

}  // namespace crn
