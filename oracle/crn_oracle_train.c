/* crn_oracle_train.c — TEST INFRASTRUCTURE ONLY (see crn_oracle.c's header): CPU twin of the device
 * trainer (crn_ann_train_device, csrc/crn_train.hip).
 *
 * The reference ships trained weights (CE_Predictive_Node.cpp:78-120) and the forward pass that uses
 * them (.cpp:214-235) but no training code (`Data Generation/TODO.md` is a note).  The trainer is this
 * build's own: classic backpropagation for the reference's exact network shape — 4 inputs, 5 sigmoid
 * hidden units, 3 sigmoid outputs, bias weights in row 0, squared-error loss — as full-batch gradient
 * descent with momentum.  The twin restates the device kernel's arithmetic *in the same order*
 * (per-lane strided partial sums, xor-butterfly over 64 lanes, waves in ascending order), so the two
 * differ only through exp(): tests compare weights to 1e-6 and decisions exactly.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/crn_sense.h"

#define NP 43            /* 5*5 input->hidden (incl. bias row) + 6*3 hidden->output (incl. bias row) */
#define THREADS 512
#define WAVES (THREADS / 64)

static uint64_t mix64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

/* parameter vector layout: w[(i*5 + (j-1))] = W_IH[i][j], i = 0..4, j = 1..5;
 *                          w[25 + j*3 + (k-1)] = W_HO[j][k], j = 0..5, k = 1..3 */
static double sigmoid(double s) { return 1.0 / (1.0 + exp(-s)); }

/* sum of v[0..THREADS) in the device's order */
static double block_sum(const double *v) {
  double total = 0.0;
  for (int w = 0; w < WAVES; w++) {
    double a[64], b[64];
    memcpy(a, v + 64 * w, sizeof(a));
    for (int off = 32; off > 0; off >>= 1) {
      for (int l = 0; l < 64; l++) b[l] = a[l] + a[l ^ off];
      memcpy(a, b, sizeof(a));
    }
    total += a[0];
  }
  return total;
}

static void train_one(const crn_train_cfg *tc, const float *feat, const int32_t *label, int64_t n,
                      const double gain[4], uint64_t seed, double *w, double *loss_out) {
  double dw[NP] = {0};
  for (int q = 0; q < NP; q++)
    w[q] = (double)(mix64(seed ^ mix64(0xBEEF00ull + (uint64_t)q)) >> 11) * (1.0 / 9007199254740992.0) - 0.5;
  double *part = (double *)malloc(sizeof(double) * THREADS * (NP + 1));
  double loss = 0.0;
  for (int it = 0; it <= tc->iterations; it++) {  /* the last pass only evaluates the loss */
    memset(part, 0, sizeof(double) * THREADS * (NP + 1));
    for (int t = 0; t < THREADS; t++) {
      double *g = part + (size_t)t * (NP + 1);
      for (int64_t s = t; s < n; s += THREADS) {
        double x[5], hid[6], out[4], d_o[4], d_h[6];
        x[0] = 1.0;
        for (int i = 0; i < 4; i++) x[i + 1] = gain[i] * (double)feat[4 * s + i];
        hid[0] = 1.0;
        for (int j = 1; j <= 5; j++) {
          double a = w[0 * 5 + (j - 1)];
          for (int i = 1; i <= 4; i++) a += x[i] * w[i * 5 + (j - 1)];
          hid[j] = sigmoid(a);
        }
        for (int k = 1; k <= 3; k++) {
          double a = w[25 + 0 * 3 + (k - 1)];
          for (int j = 1; j <= 5; j++) a += hid[j] * w[25 + j * 3 + (k - 1)];
          out[k] = sigmoid(a);
          const double target = label[s] == k ? 1.0 : 0.0;
          const double err = target - out[k];
          g[NP] += 0.5 * err * err;
          d_o[k] = err * out[k] * (1.0 - out[k]);
        }
        for (int j = 1; j <= 5; j++) {
          double a = 0.0;
          for (int k = 1; k <= 3; k++) a += w[25 + j * 3 + (k - 1)] * d_o[k];
          d_h[j] = a * hid[j] * (1.0 - hid[j]);
        }
        for (int i = 0; i <= 4; i++)
          for (int j = 1; j <= 5; j++) g[i * 5 + (j - 1)] += x[i] * d_h[j];
        for (int j = 0; j <= 5; j++)
          for (int k = 1; k <= 3; k++) g[25 + j * 3 + (k - 1)] += hid[j] * d_o[k];
      }
    }
    double col[THREADS];
    for (int t = 0; t < THREADS; t++) col[t] = part[(size_t)t * (NP + 1) + NP];
    loss = block_sum(col) / (double)n;
    if (it == tc->iterations) break;
    for (int q = 0; q < NP; q++) {
      for (int t = 0; t < THREADS; t++) col[t] = part[(size_t)t * (NP + 1) + q];
      const double grad = block_sum(col) / (double)n;
      dw[q] = (double)tc->eta * grad + (double)tc->alpha * dw[q];
      w[q] += dw[q];
    }
  }
  free(part);
  *loss_out = loss;
}

__attribute__((visibility("default")))
int crn_oracle_ann_train(const crn_train_cfg *tc, const float *feat, const int32_t *label, int64_t n,
                         double w_ih[5][6], double w_ho[6][4], double *final_loss) {
  if (!tc || !feat || !label || n < 1 || tc->iterations < 0 || tc->restarts < 1) return -1;
  double gain[4] = {1.0, 1.0, 1.0, 1.0};
  if (tc->normalise) {
    for (int i = 0; i < 4; i++) {
      double col[THREADS] = {0};
      for (int t = 0; t < THREADS; t++)
        for (int64_t s = t; s < n; s += THREADS) col[t] += (double)feat[4 * s + i];
      const double sum = block_sum(col);
      gain[i] = sum > 0.0 ? (double)n / sum : 1.0;
    }
  }
  double best[NP], best_loss = INFINITY;
  for (int r = 0; r < tc->restarts; r++) {
    double w[NP], loss;
    train_one(tc, feat, label, n, gain, tc->seed + 0x1000003ull * (uint64_t)r, w, &loss);
    if (loss < best_loss) {
      best_loss = loss;
      memcpy(best, w, sizeof(best));
    }
  }
  memset(w_ih, 0, sizeof(double) * 30);
  memset(w_ho, 0, sizeof(double) * 24);
  for (int i = 0; i <= 4; i++)
    for (int j = 1; j <= 5; j++) w_ih[i][j] = best[i * 5 + (j - 1)] * (i >= 1 ? gain[i - 1] : 1.0);  /* gains folded in */
  for (int j = 0; j <= 5; j++)
    for (int k = 1; k <= 3; k++) w_ho[j][k] = best[25 + j * 3 + (k - 1)];
  if (final_loss) *final_loss = best_loss;
  return 0;
}
