/* capi_demo.c -- libgfft.so from plain C, no Python, no torch: the drop-in boundary of
 * include/gfft.h used the way mpi4py-fft's Cython layer uses fftw_planxfftn + fftw_execute_dft
 * (mpi4py_fft/fftw/fftw_xfftn.pyx:109-157, 291-292).
 *
 *   gcc -O2 -Iinclude examples/capi_demo.c -o /tmp/capi_demo -L mpi4py-fft_amd -lgfft -lm \
 *       -Wl,-rpath,$PWD/mpi4py-fft_amd
 *
 * Plans a 3-D r2c transform of a (24, 20, 18) array and a batched 1-D c2c of length 1000 on
 * device memory obtained from gfft_malloc, executes them, and checks the results against the
 * DFT definition evaluated on the host.  Exit code 0 = parity. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "gfft.h"

#define CHECK(call)                                                                      \
  do {                                                                                   \
    int rc_ = (call);                                                                    \
    if (rc_) {                                                                           \
      fprintf(stderr, "%s failed: %s (%s)\n", #call, gfft_strerror(rc_), gfft_last_error()); \
      return 2;                                                                          \
    }                                                                                    \
  } while (0)

static double frand(unsigned *s) {
  *s = *s * 1664525u + 1013904223u;
  return (double)(*s >> 8) / 16777216.0 - 0.5;
}

int main(void) {
  const double PI2 = 6.283185307179586476925286766559;
  int ndev = 0;
  CHECK(gfft_device_count(&ndev));
  char name[128];
  CHECK(gfft_device_name(0, name, sizeof name));
  printf("libgfft %d on %s\n", gfft_version(), name);

  /* ---- batched 1-D c2c, n = 1000, 4 lines ------------------------------------------------ */
  {
    const int64_t shape[2] = {4, 1000};
    const int axes[1] = {1};
    const size_t count = 4 * 1000;
    double *h = malloc(count * 16), *out = malloc(count * 16);
    unsigned seed = 1;
    for (size_t i = 0; i < 2 * count; i++) h[i] = frand(&seed);
    void *din, *dout;
    CHECK(gfft_malloc(&din, count * 16));
    CHECK(gfft_malloc(&dout, count * 16));
    CHECK(gfft_memcpy_h2d(din, h, count * 16, NULL));
    gfft_plan plan;
    CHECK(gfft_plan_create(&plan, 2, shape, shape, 1, axes, GFFT_C2C_FORWARD, GFFT_F64));
    char desc[512];
    CHECK(gfft_plan_describe(plan, desc, sizeof desc));
    printf("%s", desc);
    CHECK(gfft_execute(plan, din, dout, 1.0, NULL));
    CHECK(gfft_memcpy_d2h(out, dout, count * 16, NULL));
    CHECK(gfft_stream_synchronize(NULL));
    double maxerr = 0, maxref = 0;
    for (int b = 0; b < 4; b++)
      for (int k = 0; k < 1000; k += 37) {
        double re = 0, im = 0;
        for (int j = 0; j < 1000; j++) {
          const double a = -PI2 * (double)((j * k) % 1000) / 1000.0, c = cos(a), s = sin(a);
          const double xr = h[2 * (b * 1000 + j)], xi = h[2 * (b * 1000 + j) + 1];
          re += xr * c - xi * s;
          im += xr * s + xi * c;
        }
        const double er = fabs(re - out[2 * (b * 1000 + k)]), ei = fabs(im - out[2 * (b * 1000 + k) + 1]);
        if (er > maxerr) maxerr = er;
        if (ei > maxerr) maxerr = ei;
        if (fabs(re) > maxref) maxref = fabs(re);
      }
    printf("1-D c2c n=1000: max |err| = %.3e (max |ref| = %.3e)\n", maxerr, maxref);
    if (maxerr > 2e-10 * maxref) return 1;
    CHECK(gfft_plan_destroy(plan));
    CHECK(gfft_free(din));
    CHECK(gfft_free(dout));
    free(h);
    free(out);
  }

  /* ---- 3-D r2c (24, 20, 18) -> (24, 20, 10), then c2r back --------------------------------- */
  {
    const int64_t sin_[3] = {24, 20, 18}, sout[3] = {24, 20, 10};
    const int axes[3] = {0, 1, 2};
    const size_t nreal = 24 * 20 * 18, ncplx = 24 * 20 * 10;
    double *h = malloc(nreal * 8), *back = malloc(nreal * 8), *spec = malloc(ncplx * 16);
    unsigned seed = 7;
    for (size_t i = 0; i < nreal; i++) h[i] = frand(&seed);
    void *dreal, *dspec, *dback;
    CHECK(gfft_malloc(&dreal, nreal * 8));
    CHECK(gfft_malloc(&dspec, ncplx * 16));
    CHECK(gfft_malloc(&dback, nreal * 8));
    CHECK(gfft_memcpy_h2d(dreal, h, nreal * 8, NULL));
    gfft_plan fwd, bck;
    CHECK(gfft_plan_create(&fwd, 3, sin_, sout, 3, axes, GFFT_R2C, GFFT_F64));
    CHECK(gfft_plan_create(&bck, 3, sout, sin_, 3, axes, GFFT_C2R, GFFT_F64));
    CHECK(gfft_execute(fwd, dreal, dspec, 1.0 / (double)nreal, NULL));   /* forward normalised, as libfft.py:412 */
    CHECK(gfft_execute(bck, dspec, dback, 1.0, NULL));
    CHECK(gfft_memcpy_d2h(spec, dspec, ncplx * 16, NULL));
    CHECK(gfft_memcpy_d2h(back, dback, nreal * 8, NULL));
    CHECK(gfft_stream_synchronize(NULL));
    /* spot-check a few spectral entries against the definition */
    double maxerr = 0;
    const int ks[4][3] = {{0, 0, 0}, {1, 2, 3}, {23, 19, 9}, {12, 10, 5}};
    for (int t = 0; t < 4; t++) {
      double re = 0, im = 0;
      for (int a = 0; a < 24; a++)
        for (int b = 0; b < 20; b++)
          for (int c = 0; c < 18; c++) {
            const double ph = -PI2 * ((double)(a * ks[t][0]) / 24 + (double)(b * ks[t][1]) / 20 + (double)(c * ks[t][2]) / 18);
            re += h[(a * 20 + b) * 18 + c] * cos(ph);
            im += h[(a * 20 + b) * 18 + c] * sin(ph);
          }
      const size_t o = ((size_t)ks[t][0] * 20 + ks[t][1]) * 10 + ks[t][2];
      const double er = fabs(re / nreal - spec[2 * o]), ei = fabs(im / nreal - spec[2 * o + 1]);
      if (er > maxerr) maxerr = er;
      if (ei > maxerr) maxerr = ei;
    }
    double rt = 0;
    for (size_t i = 0; i < nreal; i++)
      if (fabs(back[i] - h[i]) > rt) rt = fabs(back[i] - h[i]);
    printf("3-D r2c (24,20,18): spectral max |err| = %.3e, round trip max |err| = %.3e\n", maxerr, rt);
    if (maxerr > 1e-14 || rt > 1e-13) return 1;
    CHECK(gfft_plan_destroy(fwd));
    CHECK(gfft_plan_destroy(bck));
    CHECK(gfft_free(dreal));
    CHECK(gfft_free(dspec));
    CHECK(gfft_free(dback));
    free(h);
    free(back);
    free(spec);
  }
  /* 4. the exchange entries on a communicator of ONE rank (a plain C host has no launcher here;
   *    with MPI, rank 0 would MPI_Bcast the id): RCCL is bound at run time, the communicator is
   *    created and split by the real library, and an uneven all-to-all(v) to oneself plus a strided
   *    guru plan run on a stream the host owns. */
  {
    char info[256], id[GFFT_UNIQUE_ID_BYTES];
    int rc = gfft_rccl_info(info, sizeof info);
    if (rc == GFFT_ERR_UNSUPPORTED) {
      printf("exchange: no RCCL library on this host (%s) -- skipped\n", gfft_exchange_last_error());
    } else {
      CHECK(rc);
      printf("exchange: RCCL bound from %s\n", info);
      gfft_comm world = NULL, line = NULL;
      CHECK(gfft_comm_get_unique_id(id));
      CHECK(gfft_comm_create(&world, id, 1, 0));
      CHECK(gfft_comm_split(world, 0, 0, &line));
      int r = -1, n = -1;
      CHECK(gfft_comm_rank(line, &r, &n));
      if (r != 0 || n != 1) return 1;
      void *stream = NULL, *ev = NULL;
      CHECK(gfft_stream_create(&stream));
      CHECK(gfft_event_create_untimed(&ev));
      const int64_t m = 1000;
      double *hs = (double *)malloc(m * sizeof(double)), *hr = (double *)malloc(m * sizeof(double));
      for (int64_t i = 0; i < m; i++) hs[i] = (double)i;
      void *ds, *dr;
      CHECK(gfft_malloc(&ds, m * sizeof(double)));
      CHECK(gfft_malloc(&dr, m * sizeof(double)));
      CHECK(gfft_memcpy_h2d(ds, hs, m * sizeof(double), NULL));
      CHECK(gfft_stream_synchronize(NULL));
      CHECK(gfft_event_record(ev, NULL));
      CHECK(gfft_stream_wait_event(stream, ev));
      const int64_t cnt[1] = {700}, sdis[1] = {100}, rdis[1] = {300};
      CHECK(gfft_alltoallv(line, ds, cnt, sdis, dr, cnt, rdis, (int)sizeof(double), stream));
      CHECK(gfft_memcpy_d2h(hr, dr, m * sizeof(double), stream));
      CHECK(gfft_stream_synchronize(stream));
      for (int64_t i = 0; i < 700; i++)
        if (hr[300 + i] != hs[100 + i]) { printf("alltoallv mismatch at %lld\n", (long long)i); return 1; }
      /* guru plan: 64 lines of length 256 stored with stride 64 (a strided axis), out of place */
      gfft_plan g = NULL;
      gfft_iodim dim = {256, 64, 64}, hm[1] = {{64, 1, 1}};
      CHECK(gfft_plan_create_guru(&g, GFFT_F64, GFFT_C2C_FORWARD, &dim, 1, hm, 1, 0, 1, 0));
      const size_t nel = 256 * 64;
      double *hx = (double *)calloc(2 * nel, sizeof(double)), *hy = (double *)malloc(2 * nel * sizeof(double));
      for (int e = 0; e < 256; e++) hx[2 * (e * 64 + 5)] = cos(PI2 * 3 * e / 256.0), hx[2 * (e * 64 + 5) + 1] = sin(PI2 * 3 * e / 256.0);
      void *dx, *dy;
      CHECK(gfft_malloc(&dx, 2 * nel * sizeof(double)));
      CHECK(gfft_malloc(&dy, 2 * nel * sizeof(double)));
      CHECK(gfft_memcpy_h2d(dx, hx, 2 * nel * sizeof(double), stream));
      CHECK(gfft_execute(g, dx, dy, 1.0 / 256, stream));
      CHECK(gfft_memcpy_d2h(hy, dy, 2 * nel * sizeof(double), stream));
      CHECK(gfft_stream_synchronize(stream));
      /* column 5 held exp(+2 pi i 3 e / 256): all of it lands in bin 3 */
      double worst = 0;
      for (int k = 0; k < 256; k++) {
        const double want = k == 3 ? 1.0 : 0.0;
        const double er = fabs(hy[2 * (k * 64 + 5)] - want), ei = fabs(hy[2 * (k * 64 + 5) + 1]);
        if (er > worst) worst = er;
        if (ei > worst) worst = ei;
      }
      printf("guru plan (256 along a strided axis): max |err| = %.3e\n", worst);
      if (worst > 1e-14) return 1;
      CHECK(gfft_plan_destroy(g));
      CHECK(gfft_free(ds)); CHECK(gfft_free(dr)); CHECK(gfft_free(dx)); CHECK(gfft_free(dy));
      CHECK(gfft_event_destroy(ev));
      CHECK(gfft_stream_destroy(stream));
      CHECK(gfft_comm_destroy(line));
      CHECK(gfft_comm_destroy(world));
      free(hs); free(hr); free(hx); free(hy);
    }
  }
  printf("capi_demo OK\n");
  return 0;
}
