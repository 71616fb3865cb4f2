// membw_power.hip — what a streamed byte costs in watts, by load form: runs ONE streaming-read configuration over
// random data for ~3 s (so that rocm-smi, polled from outside, sees steady power) and prints its rate.
//   membw_power <aux> <width: 2 | 4 dwords per lane> [zeros]
// aux: gfx950 cache-policy bits of the buffer load (0 default, 1 sc0, 2 nt, 16 sc1).  Not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int v2u __attribute__((ext_vector_type(2)));
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
__global__ void fill(unsigned *p, size_t n, int zeros) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned long long z = i * 0x9E3779B97F4A7C15ull + 0x1234567ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    // a small fp32 value with random mantissa and sign, like noise samples
    p[i] = zeros ? 0u : ((unsigned)(z >> 32) & 0x807FFFFFu) | 0x3A000000u;
  }
}
template <int AUX, int UNROLL>
__global__ __launch_bounds__(256) void rd2(const float *p, unsigned bytes, float *out) {
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p), 0, (int)bytes, 0x00020000);
  float acc = 0.f;
  const unsigned stride = gridDim.x * 256u * UNROLL * 8u;
  for (unsigned off = (blockIdx.x * 256u * UNROLL + threadIdx.x) * 8u; off + (UNROLL - 1) * 2048u < bytes; off += stride) {
    v2u v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; u++) v[u] = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)off, u * 2048, AUX);
#pragma unroll
    for (int u = 0; u < UNROLL; u++) acc += __uint_as_float(v[u].x) + __uint_as_float(v[u].y);
  }
  if (acc == 12345.678f) out[0] = acc;
}
template <int AUX, int UNROLL>
__global__ __launch_bounds__(256) void rd4(const float *p, unsigned bytes, float *out) {
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p), 0, (int)bytes, 0x00020000);
  float acc = 0.f;
  const unsigned stride = gridDim.x * 256u * UNROLL * 16u;
  for (unsigned off = (blockIdx.x * 256u * UNROLL + threadIdx.x) * 16u; off + (UNROLL - 1) * 4096u < bytes; off += stride) {
    v4u v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; u++) v[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, u * 4096, AUX);
#pragma unroll
    for (int u = 0; u < UNROLL; u++) acc += __uint_as_float(v[u].x) + __uint_as_float(v[u].y) + __uint_as_float(v[u].z) + __uint_as_float(v[u].w);
  }
  if (acc == 12345.678f) out[0] = acc;
}
template <int AUX>
void go(int width, const float *d, unsigned bytes, float *out) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  auto launch = [&] {
    if (width == 2) hipLaunchKernelGGL((rd2<AUX, 16>), dim3(1024), dim3(256), 0, 0, d, bytes, out);
    else hipLaunchKernelGGL((rd4<AUX, 8>), dim3(1024), dim3(256), 0, 0, d, bytes, out);
  };
  for (int w = 0; w < 50; w++) launch();
  hipEventRecord(a);
  const int reps = 9000;   // ~3 s
  for (int r = 0; r < reps; r++) launch();
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  printf("aux %d width %d dwords: %.0f GB/s\n", AUX, width, (double)bytes * reps / (ms * 1e-3) / 1e9);
}
int main(int argc, char **argv) {
  const int aux = argc > 1 ? atoi(argv[1]) : 2, width = argc > 2 ? atoi(argv[2]) : 2, zeros = argc > 3;
  const unsigned bytes = 2348810240u;
  float *d, *out; hipMalloc(&d, bytes); hipMalloc(&out, 4);
  hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, (unsigned *)d, (size_t)bytes / 4, zeros);
  hipDeviceSynchronize();
  switch (aux) {
    case 0: go<0>(width, d, bytes, out); break;
    case 1: go<1>(width, d, bytes, out); break;
    case 2: go<2>(width, d, bytes, out); break;
    case 16: go<16>(width, d, bytes, out); break;
    case 18: go<18>(width, d, bytes, out); break;
    default: printf("unsupported aux\n");
  }
  return 0;
}
