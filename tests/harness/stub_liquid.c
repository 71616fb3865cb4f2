/* stub_liquid.c — TEST INFRASTRUCTURE.  Plays "the library after libcrnliquidfft in the link order"
 * (liquid-dsp in a CRTS build): its own fft_create_plan / fft_execute / fft_destroy_plan with a private plan
 * struct (an O(n^2) double-precision DFT, either direction), plus a function that — like liquid's
 * ofdmflexframegen_create, called from the ECR constructor (reference:
 * src/extensible_cognitive_radio.cpp:113) — creates a BACKWARD plan of the subcarrier count through the
 * public symbol, i.e. through whichever definition comes first in the search order. */
#include <complex.h>
#include <math.h>
#include <stdlib.h>

typedef float _Complex lfc;
struct stub_plan {          /* deliberately NOT the shim's layout */
  double magic;
  int dir;
  unsigned n;
  lfc *x, *y;
};
typedef struct stub_plan *fftplan;

#define VIS __attribute__((visibility("default")))
static int g_created, g_executed, g_destroyed;

VIS fftplan fft_create_plan(unsigned int n, lfc *x, lfc *y, int dir, int flags) {
  (void)flags;
  struct stub_plan *p = (struct stub_plan *)malloc(sizeof(*p));
  p->magic = 1234.5;
  p->dir = dir;
  p->n = n;
  p->x = x;
  p->y = y;
  g_created++;
  return p;
}

VIS void fft_execute(fftplan p) {
  if (p->magic != 1234.5) abort(); /* somebody else's plan was routed here */
  const double sgn = p->dir > 0 ? -1.0 : 1.0;
  for (unsigned k = 0; k < p->n; k++) {
    double complex s = 0;
    for (unsigned i = 0; i < p->n; i++)
      s += (double complex)p->x[i] * cexp(sgn * 2.0 * M_PI * I * (double)(((unsigned long)k * i) % p->n) / (double)p->n);
    p->y[k] = (lfc)s;
  }
  g_executed++;
}

VIS void fft_destroy_plan(fftplan p) {
  if (p->magic != 1234.5) abort();
  g_destroyed++;
  free(p);
}

/* "liquid-internal" user of the public symbols: an M-subcarrier backward plan, executed once. */
VIS int stub_ofdm_like_create(unsigned int m, lfc *freq, lfc *time) {
  fftplan p = fft_create_plan(m, freq, time, -1, 0);
  fft_execute(p);
  fft_destroy_plan(p);
  return 0;
}

VIS void stub_counts(int *created, int *executed, int *destroyed) {
  *created = g_created;
  *executed = g_executed;
  *destroyed = g_destroyed;
}
