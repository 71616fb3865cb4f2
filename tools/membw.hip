// membw.hip — streaming-read calibration for MI355X: what HBM read rate does a plain
// grid-stride reduction reach with 8-byte and 16-byte per-lane loads?  (Not part of the product.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <typename V, int UNROLL, bool NT>
__global__ __launch_bounds__(256) void rd(const V *p, size_t n, float *out) {
  size_t i = (size_t)blockIdx.x * blockDim.x * UNROLL + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x * UNROLL;
  float acc = 0.f;
  for (; i + (UNROLL - 1) * blockDim.x < n; i += stride) {
    V v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; u++) {
      if constexpr (NT) v[u] = __builtin_nontemporal_load(p + i + u * blockDim.x);
      else v[u] = p[i + u * blockDim.x];
    }
#pragma unroll
    for (int u = 0; u < UNROLL; u++) acc += v[u][0] + v[u][sizeof(V) / 4 - 1];
  }
  if (acc == 12345.678f) out[0] = acc;
}

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

template <typename V, int UNROLL, bool NT>
float run(const void *d, size_t bytes, float *out, int blocks) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  const size_t n = bytes / sizeof(V);
  for (int w = 0; w < 2; w++) hipLaunchKernelGGL((rd<V, UNROLL, NT>), dim3(blocks), dim3(256), 0, 0, (const V *)d, n, out);
  hipEventRecord(a);
  const int reps = 10;
  for (int r = 0; r < reps; r++) hipLaunchKernelGGL((rd<V, UNROLL, NT>), dim3(blocks), dim3(256), 0, 0, (const V *)d, n, out);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return bytes * (float)reps / (ms * 1e-3f) / 1e9f;
}

int main() {
  const size_t bytes = (size_t)2348810240;  // same footprint as the bench batch
  void *d; float *out;
  CK(hipMalloc(&d, bytes)); CK(hipMalloc(&out, 4));
  CK(hipMemset(d, 1, bytes));
  // bytes in flight per CU = blocks/CU * 256 threads * UNROLL * bytes per load
  for (int blocks : {256, 512, 768, 1024}) {
    printf("blocks=%5d x2/nt: u1 %6.0f  u2 %6.0f  u4 %6.0f  u8 %6.0f  u16 %6.0f  u32 %6.0f GB/s (in flight/CU: %d,%d,%d,%d,%d,%d KiB)\n", blocks,
           run<v2f, 1, true>(d, bytes, out, blocks), run<v2f, 2, true>(d, bytes, out, blocks), run<v2f, 4, true>(d, bytes, out, blocks),
           run<v2f, 8, true>(d, bytes, out, blocks), run<v2f, 16, true>(d, bytes, out, blocks), run<v2f, 32, true>(d, bytes, out, blocks),
           blocks / 256 * 2, blocks / 256 * 4, blocks / 256 * 8, blocks / 256 * 16, blocks / 256 * 32, blocks / 256 * 64);
  }
  for (int blocks : {256 * 2, 256 * 8, 256 * 32}) {
    printf("blocks=%5d  x2/u4 %7.0f  x2/u16 %7.0f  x2/u16/nt %7.0f  x4/u4 %7.0f  x4/u8 %7.0f  x4/u8/nt %7.0f GB/s\n", blocks,
           run<v2f, 4, false>(d, bytes, out, blocks), run<v2f, 16, false>(d, bytes, out, blocks),
           run<v2f, 16, true>(d, bytes, out, blocks), run<v4f, 4, false>(d, bytes, out, blocks),
           run<v4f, 8, false>(d, bytes, out, blocks), run<v4f, 8, true>(d, bytes, out, blocks));
  }
  return 0;
}
