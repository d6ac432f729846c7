/* A plain-C caller of libgcmf (no Python, no torch): REGULAR grid, 3-step polynomial, host buffers.
 * Exit codes: 0 = result matches the inline C restatement of filter.py:185-210 bit for bit,
 *             3 = no GPU (GCMF_ERR_NO_DEVICE) -- what the CPU-only test expects, anything else = failure. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "gcmf.h"

#define NY 48
#define NX 64

static double lap(const double *f, int j, int i) { /* kernels.py:115-121, periodic */
  int jn = (j + 1) % NY, js = (j + NY - 1) % NY, ie = (i + 1) % NX, iw = (i + NX - 1) % NX;
  double l = -4.0 * f[j * NX + i] + f[j * NX + ie];
  l = l + f[j * NX + iw];
  l = l + f[jn * NX + i];
  l = l + f[js * NX + i];
  return l;
}

int main(void) {
  static double f[NY * NX], out[NY * NX], t0[NY * NX], t1[NY * NX], t2[NY * NX], fb[NY * NX], a[NY * NX];
  const double p[4] = {0.4, -0.3, 0.2, -0.1}; /* p(-1) = 1 */
  const int n = 3;
  const double c = 0.25;
  unsigned s = 12345u;
  for (int q = 0; q < NY * NX; ++q) { s = s * 1664525u + 1013904223u; f[q] = (double)(s >> 8) / 16777216.0; }

  gcmf_plan_desc d = {GCMF_REGULAR, GCMF_F64, NY, NX, 0, NY, 0, 0, 0, 0};
  gcmf_plan *plan = NULL;
  int st = gcmf_plan_create(&d, NULL, 0, &plan);
  if (st != GCMF_OK) {
    fprintf(stderr, "gcmf_plan_create: status %d: %s\n", st, gcmf_last_error());
    return st;
  }
  const void *in[1] = {f};
  void *o[1] = {out};
  st = gcmf_apply(plan, p, n, c, in, o, 1, 0, NULL);
  if (st != GCMF_OK) {
    fprintf(stderr, "gcmf_apply: status %d: %s\n", st, gcmf_last_error());
    return 10 + st;
  }
  gcmf_plan_destroy(plan);

  /* reference recurrence */
  for (int q = 0; q < NY * NX; ++q) t2[q] = f[q];
  for (int j = 0; j < NY; ++j)
    for (int i = 0; i < NX; ++i) { int q = j * NX + i; t1[q] = -f[q] - c * lap(f, j, i); fb[q] = p[0] * t2[q] + p[1] * t1[q]; }
  for (int k = 2; k <= n; ++k) {
    for (int j = 0; j < NY; ++j)
      for (int i = 0; i < NX; ++i) { int q = j * NX + i; a[q] = -t1[q] - c * lap(t1, j, i); }
    for (int q = 0; q < NY * NX; ++q) { t0[q] = 2 * a[q] - t2[q]; fb[q] += p[k] * t0[q]; t2[q] = t1[q]; t1[q] = t0[q]; }
  }
  int bad = 0;
  for (int q = 0; q < NY * NX; ++q) bad += (out[q] != fb[q]);
  printf("cabi_smoke: %d of %d cells differ\n", bad, NY * NX);
  return bad ? 1 : 0;
}
