/* liquid_shim_check.c — TEST INFRASTRUCTURE.  Linked as INTEGRATION.md §6 says a CRTS node would be:
 *     ... -lcrnliquidfft -lcrnsense ... -l<liquid>      (here: libstub_liquid.so stands in for liquid)
 * and then does what such a process does with liquid's FFT entry points:
 *   1. a "liquid-internal" 64-point BACKWARD plan (the ECR constructor's OFDM framing plans,
 *      reference: src/extensible_cognitive_radio.cpp:113,123) — must reach the library behind the shim;
 *   2. the same through the public symbol from application code;
 *   3. (with `gpu` as argv[1]) the sensing path's 512-point FORWARD plan — must run on the GPU, beside 1 and 2.
 * Prints the numbers the test asserts on. */
#include <complex.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef float _Complex lfc;
typedef struct fftplan_s *fftplan;
fftplan fft_create_plan(unsigned int n, lfc *x, lfc *y, int dir, int flags);
void fft_execute(fftplan p);
void fft_destroy_plan(fftplan p);
int stub_ofdm_like_create(unsigned int m, lfc *freq, lfc *time);
void stub_counts(int *created, int *executed, int *destroyed);
long crn_liquid_fft_forwarded(void);

static double idft_err(unsigned n, const lfc *x, const lfc *y) { /* y ?= sum_k x[k] e^{+j 2 pi k i / n} */
  double worst = 0;
  for (unsigned i = 0; i < n; i++) {
    double complex s = 0;
    for (unsigned k = 0; k < n; k++) s += (double complex)x[k] * cexp(2.0 * M_PI * I * (double)((k * i) % n) / n);
    if (cabs(s - y[i]) > worst) worst = cabs(s - y[i]);
  }
  return worst;
}

int main(int argc, char **argv) {
  enum { M = 64, N = 512 };
  static lfc f[M], t[M], t2[M], x[N], y[N];
  for (int k = 0; k < M; k++) f[k] = (float)cos(0.3 * k) + I * (float)sin(0.7 * k + 1);
  stub_ofdm_like_create(M, f, t);
  printf("internal_backward_err %.3g\n", idft_err(M, f, t));
  fftplan p = fft_create_plan(M, f, t2, -1, 0);
  fft_execute(p);
  fft_destroy_plan(p);
  printf("public_backward_err %.3g\n", idft_err(M, f, t2));
  if (argc > 1 && strcmp(argv[1], "gpu") == 0) {
    for (int i = 0; i < N; i++) x[i] = (float)cos(2 * M_PI * 37 * i / N) * 0.5f + I * (float)sin(0.01 * i * i);
    fftplan q = fft_create_plan(N, x, y, +1, 0);   /* CE_Predictive_Node.cpp:42-45 */
    fft_execute(q);                                /* :150 */
    double worst = 0, rms = 0;
    for (int k = 0; k < N; k++) {
      double complex s = 0;
      for (int i = 0; i < N; i++) s += (double complex)x[i] * cexp(-2.0 * M_PI * I * (double)((k * i) % N) / N);
      if (cabs(s - y[k]) > worst) worst = cabs(s - y[k]);
      rms += cabs(s) * cabs(s) / N;
    }
    fft_destroy_plan(q);
    printf("gpu_forward_rel_err %.3g\n", worst / sqrt(rms));
  }
  int c, e, d;
  stub_counts(&c, &e, &d);
  printf("next_library_plans created %d executed %d destroyed %d\n", c, e, d);
  printf("forwarded_by_shim %ld\n", crn_liquid_fft_forwarded());
  return 0;
}
