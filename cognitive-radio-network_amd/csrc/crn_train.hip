// crn_train.hip — crn_ann_train_device (include/crn_sense.h): fits the reference's 4-5-3 sigmoid
// network (CE_Predictive_Node.hpp:20-22,62-73; forward pass CE_Predictive_Node.cpp:214-235) to labelled
// features resident in HBM.  gfx950 only.
//
// One workgroup of 512 threads = one restart: the whole descent runs inside one launch, no host
// round trips.  The 43 parameters live in LDS; every iteration each lane folds its strided share of
// the samples into 43 + 1 register partials (fp64, like the reference's forward pass), the partials
// are reduced by an xor butterfly over the wave and then over the 8 waves in ascending order —
// a fixed order that oracle/crn_oracle_train.c restates, so the two differ only through exp().
#include <hip/hip_runtime.h>

#include <cmath>
#include <string>
#include <vector>

#include "../../include/crn_sense.h"
#include "crn_internal.h"

extern "C" int crn_sense_cfg_of(crn_handle *h, crn_cfg *out);

namespace crn {

constexpr int kParams = 43;  // 5x5 input->hidden incl. bias row + 6x3 hidden->output incl. bias row
constexpr int kTrainThreads = 512;
constexpr int kTrainWaves = kTrainThreads / 64;

struct TrainParams {
  const float *feat;     // [n][4]
  const int32_t *label;  // [n]
  long long n;
  unsigned long long seed;
  int iterations;
  int normalise;
  double eta, alpha;
  double *out;           // [restarts][kParams + 1 + 4]: parameters, loss, gains
};

__device__ __forceinline__ unsigned long long tmix64(unsigned long long z) {  // splitmix64 step
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

// Sum of one value per thread, same result in every thread; order: xor butterfly 32, 16, .., 1 over
// the wave, then the waves in sequence.
__device__ double block_sum(double v, double *scratch /* [kTrainWaves] */) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  const int wave = threadIdx.x >> 6;
  __syncthreads();  // scratch free
  if ((threadIdx.x & 63) == 0) scratch[wave] = v;
  __syncthreads();
  double total = 0.0;
#pragma unroll
  for (int w = 0; w < kTrainWaves; w++) total += scratch[w];
  return total;
}

__global__ __launch_bounds__(kTrainThreads) void ann_train_kernel(const TrainParams p) {
  __shared__ double w[kParams], dw[kParams], gain[4], scratch[kTrainWaves];
  const int t = threadIdx.x;
  const unsigned long long seed = p.seed + 0x1000003ull * (unsigned long long)blockIdx.x;

  // feature gains: 1 / mean_i (folded back into W_IH by the host)
  for (int i = 0; i < 4; i++) {
    double s = 0.0;
    if (p.normalise)
      for (long long k = t; k < p.n; k += kTrainThreads) s += (double)p.feat[4 * k + i];
    const double sum = block_sum(s, scratch);
    if (t == 0) gain[i] = (p.normalise && sum > 0.0) ? (double)p.n / sum : 1.0;
  }
  if (t < kParams) {
    w[t] = (double)(tmix64(seed ^ tmix64(0xBEEF00ull + (unsigned long long)t)) >> 11) * (1.0 / 9007199254740992.0) - 0.5;
    dw[t] = 0.0;
  }
  __syncthreads();

  double loss = 0.0;
  for (int it = 0; it <= p.iterations; it++) {  // the last pass only evaluates the loss
    double g[kParams + 1];
#pragma unroll
    for (int q = 0; q <= kParams; q++) g[q] = 0.0;
    double wl[kParams];
#pragma unroll
    for (int q = 0; q < kParams; q++) wl[q] = w[q];
    for (long long s = t; s < p.n; s += kTrainThreads) {
      double x[5], hid[6], d_o[4], d_h[6];
      x[0] = 1.0;
#pragma unroll
      for (int i = 0; i < 4; i++) x[i + 1] = gain[i] * (double)p.feat[4 * s + i];
      const int lab = p.label[s];
      hid[0] = 1.0;
#pragma unroll
      for (int j = 1; j <= 5; j++) {
        double a = wl[0 * 5 + (j - 1)];
#pragma unroll
        for (int i = 1; i <= 4; i++) a += x[i] * wl[i * 5 + (j - 1)];
        hid[j] = 1.0 / (1.0 + exp(-a));
      }
#pragma unroll
      for (int k = 1; k <= 3; k++) {
        double a = wl[25 + 0 * 3 + (k - 1)];
#pragma unroll
        for (int j = 1; j <= 5; j++) a += hid[j] * wl[25 + j * 3 + (k - 1)];
        const double out = 1.0 / (1.0 + exp(-a));
        const double err = (lab == k ? 1.0 : 0.0) - out;
        g[kParams] += 0.5 * err * err;
        d_o[k] = err * out * (1.0 - out);
      }
#pragma unroll
      for (int j = 1; j <= 5; j++) {
        double a = 0.0;
#pragma unroll
        for (int k = 1; k <= 3; k++) a += wl[25 + j * 3 + (k - 1)] * d_o[k];
        d_h[j] = a * hid[j] * (1.0 - hid[j]);
      }
#pragma unroll
      for (int i = 0; i <= 4; i++)
#pragma unroll
        for (int j = 1; j <= 5; j++) g[i * 5 + (j - 1)] += x[i] * d_h[j];
#pragma unroll
      for (int j = 0; j <= 5; j++)
#pragma unroll
        for (int k = 1; k <= 3; k++) g[25 + j * 3 + (k - 1)] += hid[j] * d_o[k];
    }
    loss = block_sum(g[kParams], scratch) / (double)p.n;
    if (it == p.iterations) break;
#pragma unroll
    for (int q = 0; q < kParams; q++) {
      const double grad = block_sum(g[q], scratch) / (double)p.n;
      if (t == 0) {
        dw[q] = p.eta * grad + p.alpha * dw[q];
        w[q] += dw[q];
      }
    }
    __syncthreads();
  }
  double *out = p.out + (size_t)blockIdx.x * (kParams + 1 + 4);
  if (t < kParams) out[t] = w[t];
  if (t == 0) out[kParams] = loss;
  if (t < 4) out[kParams + 1 + t] = gain[t];
}

}  // namespace crn

#define HIP_TRY(expr)                                                                          \
  do {                                                                                         \
    hipError_t _e = (expr);                                                                    \
    if (_e != hipSuccess)                                                                      \
      return crn::fail(CRN_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(_e));     \
  } while (0)

extern "C" int crn_ann_train_device(crn_handle *h, const crn_train_cfg *tc, const float *d_features,
                                    const int32_t *d_labels, int64_t n, double w_ih[5][6], double w_ho[6][4],
                                    double *final_loss, void *stream) {
  if (!h || !tc || !d_features || !d_labels || !w_ih || !w_ho) return crn::fail(CRN_ERR_ARG, "null argument");
  if (n < 1) return crn::fail(CRN_ERR_ARG, "no samples");
  if (tc->iterations < 0 || tc->restarts < 1 || tc->restarts > 4096) return crn::fail(CRN_ERR_ARG, "bad iterations / restarts");
  if (!(tc->eta > 0.f) || tc->alpha < 0.f || tc->alpha >= 1.f) return crn::fail(CRN_ERR_ARG, "bad eta / alpha");
  crn_cfg cfg;
  if (int rc = crn_sense_cfg_of(h, &cfg)) return rc;
  if (cfg.n_bands != 4) return crn::fail(CRN_ERR_ARG, "the 4-5-3 network takes the four features {NF, CH1, CH2, CH3}");
  HIP_TRY(hipSetDevice(cfg.device));
  const size_t per = crn::kParams + 1 + 4;
  double *d_out = nullptr;
  HIP_TRY(hipMalloc(&d_out, sizeof(double) * per * tc->restarts));
  crn::TrainParams p{};
  p.feat = d_features;
  p.label = d_labels;
  p.n = n;
  p.seed = tc->seed;
  p.iterations = tc->iterations;
  p.normalise = tc->normalise ? 1 : 0;
  p.eta = (double)tc->eta;
  p.alpha = (double)tc->alpha;
  p.out = d_out;
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(crn::ann_train_kernel, dim3(tc->restarts), dim3(crn::kTrainThreads), 0, st, p);
  hipError_t e = hipGetLastError();
  std::vector<double> host(per * tc->restarts);
  if (e == hipSuccess) e = hipMemcpyAsync(host.data(), d_out, sizeof(double) * host.size(), hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  (void)hipFree(d_out);
  if (e != hipSuccess) return crn::fail(CRN_ERR_DEVICE, std::string("ann_train: ") + hipGetErrorString(e));
  int best = 0;
  for (int r = 1; r < tc->restarts; r++)
    if (host[r * per + crn::kParams] < host[best * per + crn::kParams]) best = r;  // ties: lowest index
  const double *w = &host[best * per];
  const double *gain = w + crn::kParams + 1;
  for (int i = 0; i < 5; i++)
    for (int j = 0; j < 6; j++) w_ih[i][j] = 0.0;
  for (int j = 0; j < 6; j++)
    for (int k = 0; k < 4; k++) w_ho[j][k] = 0.0;
  for (int i = 0; i <= 4; i++)
    for (int j = 1; j <= 5; j++) w_ih[i][j] = w[i * 5 + (j - 1)] * (i >= 1 ? gain[i - 1] : 1.0);  // gains folded in
  for (int j = 0; j <= 5; j++)
    for (int k = 1; k <= 3; k++) w_ho[j][k] = w[25 + j * 3 + (k - 1)];
  if (final_loss) *final_loss = w[crn::kParams];
  return CRN_OK;
}
