// membw_policy.hip — streaming-read rate of buffer_load_dwordx2 by cache-policy bits (gfx950:
// aux bit0 = sc0, bit1 = nt, bit4 = sc1).  Not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int v2u __attribute__((ext_vector_type(2)));
template <int AUX, int UNROLL>
__global__ __launch_bounds__(256) void rd(const float *p, unsigned bytes, float *out) {
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
template <int AUX>
float run(const float *d, unsigned bytes, float *out) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int w = 0; w < 3; w++) hipLaunchKernelGGL((rd<AUX, 16>), dim3(1024), dim3(256), 0, 0, d, bytes, out);
  hipEventRecord(a);
  for (int r = 0; r < 20; r++) hipLaunchKernelGGL((rd<AUX, 16>), dim3(1024), dim3(256), 0, 0, d, bytes, out);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return bytes * 20.0f / (ms * 1e-3f) / 1e9f;
}
int main() {
  const unsigned bytes = 2348810240u;
  float *d, *out; hipMalloc(&d, bytes); hipMalloc(&out, 4); hipMemset(d, 1, bytes);
  for (int rep = 0; rep < 2; rep++)
    printf("aux 0:%6.0f  sc0:%6.0f  nt:%6.0f  sc0+nt:%6.0f  sc1:%6.0f  sc0+sc1:%6.0f  nt+sc1:%6.0f  sc0+nt+sc1:%6.0f GB/s\n",
           run<0>(d, bytes, out), run<1>(d, bytes, out), run<2>(d, bytes, out), run<3>(d, bytes, out), run<16>(d, bytes, out),
           run<17>(d, bytes, out), run<18>(d, bytes, out), run<19>(d, bytes, out));
  return 0;
}
