// valu_rate.hip — cycles per wave-instruction of f32 VALU forms on gfx950, by waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
#define REP16(x) x x x x x x x x x x x x x x x x
template <int MODE>
__global__ __launch_bounds__(1024) void k(float *out, unsigned long long *cyc, int iters) {
  v2f a0 = {1.f + threadIdx.x, 2.f}, a1 = {3.f, 4.f}, a2 = {5.f, 6.f}, a3 = {7.f, 8.f};
  v2f a4 = {1.5f, 2.5f}, a5 = {3.5f, 4.5f}, a6 = {5.5f, 6.5f}, a7 = {7.5f, 8.5f};
  v2f b = {1.0001f, 0.9999f}, c = {1e-3f, 2e-3f};
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; i++) {
    if constexpr (MODE == 0) {  // v_pk_fma_f32, 8 independent chains
      REP16(asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                   "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));)
    } else if constexpr (MODE == 1) {  // v_fma_f32
      REP16(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                   "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                   : "+v"(a0.x), "+v"(a1.x), "+v"(a2.x), "+v"(a3.x), "+v"(a4.x), "+v"(a5.x), "+v"(a6.x), "+v"(a7.x) : "v"(b.x), "v"(c.x));)
    } else if constexpr (MODE == 2) {  // v_pk_add_f32 with op_sel / neg modifiers
      REP16(asm volatile("v_pk_add_f32 %0, %0, %8 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n v_pk_add_f32 %1, %1, %8 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n"
                   "v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %8\n v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));)
    } else if constexpr (MODE == 3) {  // v_add_f32
      REP16(asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                   "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n"
                   : "+v"(a0.x), "+v"(a1.x), "+v"(a2.x), "+v"(a3.x), "+v"(a4.x), "+v"(a5.x), "+v"(a6.x), "+v"(a7.x) : "v"(b.x), "v"(c.x));)
    } else if constexpr (MODE == 4) {  // v_pk_mul_f32
      REP16(asm volatile("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n"
                   "v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));)
    } else if constexpr (MODE == 5) {  // v_mov_b32
      REP16(asm volatile("v_mov_b32 %0, %8\n v_mov_b32 %1, %8\n v_mov_b32 %2, %8\n v_mov_b32 %3, %8\n"
                   "v_mov_b32 %4, %8\n v_mov_b32 %5, %8\n v_mov_b32 %6, %8\n v_mov_b32 %7, %8\n"
                   : "+v"(a0.x), "+v"(a1.x), "+v"(a2.x), "+v"(a3.x), "+v"(a4.x), "+v"(a5.x), "+v"(a6.x), "+v"(a7.x) : "v"(b.x), "v"(c.x));)
    } else {  // 6: v_pk_fma with an SGPR-pair operand
      REP16(asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                   "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(b), "v"(c));)
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  v2f s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
  if (s.x == 123.456f) out[0] = s.y;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int MODE>
void run(const char *name, float *out, unsigned long long *cyc) {
  const int iters = 4000;
  for (int wps : {1, 2, 3, 4}) {  // waves per SIMD = block threads / 256, one block per CU
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256 * wps), 0, 0, out, cyc, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256 * wps), 0, 0, out, cyc, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    double per_inst = (double)c / (iters * 128.0);
    double ns_simd = ms * 1e6 / (iters * 128.0 * wps);  // wall ns per wave-instruction per SIMD
    printf("%-28s waves/SIMD=%d  memtime ticks/inst(one wave)=%.3f  wall: %.3f ns per inst per SIMD = %.2f cyc @2.4GHz (kernel %.3f ms)\n",
           name, wps, per_inst, ns_simd, ns_simd * 2.4, ms);
  }
}
int main() {
  float *out; unsigned long long *cyc;
  hipMalloc(&out, 4); hipMalloc(&cyc, 8);
  run<0>("v_pk_fma_f32", out, cyc); run<1>("v_fma_f32", out, cyc); run<2>("v_pk_add_f32(op_sel,neg)", out, cyc);
  run<3>("v_add_f32", out, cyc); run<4>("v_pk_mul_f32", out, cyc); run<5>("v_mov_b32", out, cyc); run<6>("v_pk_fma_f32 (sgpr src)", out, cyc);
  return 0;
}
