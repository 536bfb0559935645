/* dft_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C restatement of what the reference's native planner asks FFTW to compute:
 * mpi4py_fft/fftw/fftw_planxfftn.c:10-77 turns (sizes, axes, kind) into FFTW guru dims --
 * row-major strides (.c:25-30), one transform dim per listed axis (.c:34-40), every other dim
 * a batch dim (.c:41-47) -- and FFTW then evaluates, per FFTW's documented definition,
 *     Y[k] = sum_j X[j] exp(-/+ 2 pi i j k / n)            (unnormalised; - forward, + backward)
 * r2c keeps k = 0..n/2 of the forward transform of real data; c2r is the backward transform of
 * the Hermitian extension (real output).  This file evaluates that definition directly,
 * O(n^2) per line in long double -- independent of any FFT algorithm -- for small shapes.
 * libfftw3 itself is absent from the image and unpinned by the reference (setup.py:64-81).
 *
 * Parity status: pinned by tests/test_oracle_golden.py::test_c_oracle_* against the fixtures
 * generated from the reference's Python (numpy backend) and the docstring KATs.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

enum { C2C_FORWARD = -1, C2C_BACKWARD = 1, R2C = -2, C2R = 2 };

static const long double PI2 = 6.283185307179586476925286766559005768L;

/* one axis pass over a complex array of shape sizes[], transform length n along `axis`;
 * in/out are interleaved long double complex with the given shapes (they may differ along axis) */
static void pass_c2c(int ndims, const int64_t *sizes, int axis, int sign, long double *buf) {
  int64_t n = sizes[axis], outer = 1, inner = 1;
  for (int i = 0; i < axis; i++) outer *= sizes[i];
  for (int i = axis + 1; i < ndims; i++) inner *= sizes[i];
  long double *tmp = malloc(sizeof(long double) * 2 * n);
  for (int64_t o = 0; o < outer; o++)
    for (int64_t i = 0; i < inner; i++) {
      long double *base = buf + 2 * (o * n * inner + i);
      for (int64_t k = 0; k < n; k++) {
        long double re = 0, im = 0;
        for (int64_t j = 0; j < n; j++) {
          long double a = sign * PI2 * (long double)((j * k) % n) / (long double)n;
          long double c = cosl(a), s = sinl(a);
          long double xr = base[2 * j * inner], xi = base[2 * j * inner + 1];
          re += xr * c - xi * s;
          im += xr * s + xi * c;
        }
        tmp[2 * k] = re;
        tmp[2 * k + 1] = im;
      }
      for (int64_t k = 0; k < n; k++) {
        base[2 * k * inner] = tmp[2 * k];
        base[2 * k * inner + 1] = tmp[2 * k + 1];
      }
    }
  free(tmp);
}

static int64_t prod(int n, const int64_t *s) {
  int64_t p = 1;
  for (int i = 0; i < n; i++) p *= s[i];
  return p;
}

/* Same argument list as fftw_planxfftn (minus flags) + execute.  `in`/`out` are double arrays:
 * real arrays hold 1 double per element, complex 2.  Returns 0, or -1 on bad arguments. */
int dft_oracle_xfftn(int ndims, const int64_t *sizes_in, const double *in, const int64_t *sizes_out,
                     double *out, int naxes, const int *axes, int kind) {
  if (ndims < 1 || naxes < 1) return -1;
  int last = axes[naxes - 1];
  /* work on the "full complex" shape: the real side's shape */
  const int64_t *full = (kind == C2R) ? sizes_out : sizes_in;
  int64_t total = prod(ndims, full);
  long double *w = calloc((size_t)total * 2, sizeof(long double));
  if (!w) return -1;
  int64_t nl = full[last], nh = nl / 2 + 1;
  int64_t outer = 1, inner = 1;
  for (int i = 0; i < last; i++) outer *= full[i];
  for (int i = last + 1; i < ndims; i++) inner *= full[i];
  if (kind == R2C) {
    for (int64_t t = 0; t < total; t++) w[2 * t] = in[t];
  } else if (kind == C2R) {
    /* the other axes first on the half spectrum, then Hermitian extension along `last` */
    int64_t htotal = prod(ndims, sizes_in);
    long double *h = malloc(sizeof(long double) * 2 * htotal);
    for (int64_t t = 0; t < 2 * htotal; t++) h[t] = in[t];
    for (int a = 0; a < naxes - 1; a++) pass_c2c(ndims, sizes_in, axes[a], +1, h);
    for (int64_t o = 0; o < outer; o++)
      for (int64_t k = 0; k < nl; k++)
        for (int64_t i = 0; i < inner; i++) {
          int64_t kk = k < nh ? k : nl - k;
          long double re = h[2 * ((o * nh + kk) * inner + i)], im = h[2 * ((o * nh + kk) * inner + i) + 1];
          w[2 * ((o * nl + k) * inner + i)] = re;
          w[2 * ((o * nl + k) * inner + i) + 1] = k < nh ? im : -im;
        }
    free(h);
    pass_c2c(ndims, full, last, +1, w);
    for (int64_t t = 0; t < total; t++) out[t] = (double)w[2 * t];
    free(w);
    return 0;
  } else {
    for (int64_t t = 0; t < 2 * total; t++) w[t] = in[t];
  }
  int sign = (kind == C2C_BACKWARD) ? +1 : -1;
  for (int a = naxes - 1; a >= 0; a--) pass_c2c(ndims, full, axes[a], sign, w);
  if (kind == R2C) {
    for (int64_t o = 0; o < outer; o++)
      for (int64_t k = 0; k < nh; k++)
        for (int64_t i = 0; i < inner; i++) {
          out[2 * ((o * nh + k) * inner + i)] = (double)w[2 * ((o * nl + k) * inner + i)];
          out[2 * ((o * nh + k) * inner + i) + 1] = (double)w[2 * ((o * nl + k) * inner + i) + 1];
        }
  } else {
    for (int64_t t = 0; t < 2 * total; t++) out[t] = (double)w[t];
  }
  free(w);
  return 0;
}
