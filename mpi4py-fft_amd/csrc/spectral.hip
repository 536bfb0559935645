// Pointwise kernels of the pseudo-spectral caller either side of the transform path
// (the reference's examples/spectral_dns_solver.py:65-91 does these with numpy expressions, one
// temporary per operator): i K x u_hat, u x w, pressure projection + viscous term, and the
// Runge-Kutta stage update, each one pass over its operands.  HBM-bound elementwise work:
// 16 bytes per lane, grid-stride, wavenumbers from three per-axis vectors instead of three
// array-sized meshes (the reference's K is a (3, n0, n1, n2) float array: 1.5x the velocity).
#include "gfft_internal.h"

namespace gfft {

namespace {

constexpr int PS_THREADS = 256;

inline int ps_grid(int64_t count) {
  const int64_t b = (count + PS_THREADS - 1) / PS_THREADS;
  return (int)(b < 16384 ? (b > 0 ? b : 1) : 16384);
}

// out = i (K x u): out0 = i (k1 u2 - k2 u1), out1 = i (k2 u0 - k0 u2), out2 = i (k0 u1 - k1 u0)
template <typename real>
__global__ void __launch_bounds__(PS_THREADS)
ps_curl_kernel(const cx<real> *__restrict__ u, cx<real> *__restrict__ out, const real *__restrict__ k0,
               const real *__restrict__ k1, const real *__restrict__ k2, int64_t n1, int64_t n2, int64_t count) {
  for (int64_t e = (int64_t)blockIdx.x * PS_THREADS + threadIdx.x; e < count; e += (int64_t)gridDim.x * PS_THREADS) {
    const int64_t row = e / n2;
    const real kz = k2[e - row * n2];
    const int64_t i0 = row / n1;
    const real ky = k1[row - i0 * n1], kx = k0[i0];
    const cx<real> a = u[e], b = u[e + count], c = u[e + 2 * count];
    const cx<real> w0 = {ky * c.x - kz * b.x, ky * c.y - kz * b.y};
    const cx<real> w1 = {kz * a.x - kx * c.x, kz * a.y - kx * c.y};
    const cx<real> w2 = {kx * b.x - ky * a.x, kx * b.y - ky * a.y};
    out[e] = {-w0.y, w0.x};
    out[e + count] = {-w1.y, w1.x};
    out[e + 2 * count] = {-w2.y, w2.x};
  }
}

// out = a x b on real 3-vectors stored as [3][count]
template <typename real>
__global__ void __launch_bounds__(PS_THREADS)
ps_cross_kernel(const real *__restrict__ a, const real *__restrict__ b, real *__restrict__ out, int64_t count) {
  for (int64_t e = (int64_t)blockIdx.x * PS_THREADS + threadIdx.x; e < count; e += (int64_t)gridDim.x * PS_THREADS) {
    const real a0 = a[e], a1 = a[e + count], a2 = a[e + 2 * count];
    const real b0 = b[e], b1 = b[e + count], b2 = b[e + 2 * count];
    out[e] = a1 * b2 - a2 * b1;
    out[e + count] = a2 * b0 - a0 * b2;
    out[e + 2 * count] = a0 * b1 - a1 * b0;
  }
}

// p = sum_i du_i k_i / |k|^2 (|k|^2 = 0 -> 1);  du_j -= k_j p;  du_j -= nu |k|^2 u_j
template <typename real>
__global__ void __launch_bounds__(PS_THREADS)
ps_project_kernel(cx<real> *__restrict__ du, const cx<real> *__restrict__ u, const real *__restrict__ k0,
                  const real *__restrict__ k1, const real *__restrict__ k2, int64_t n1, int64_t n2,
                  int64_t count, real nu) {
  for (int64_t e = (int64_t)blockIdx.x * PS_THREADS + threadIdx.x; e < count; e += (int64_t)gridDim.x * PS_THREADS) {
    const int64_t row = e / n2;
    const real kz = k2[e - row * n2];
    const int64_t i0 = row / n1;
    const real ky = k1[row - i0 * n1], kx = k0[i0];
    const real kk = kx * kx + ky * ky + kz * kz;
    const real inv = (real)1 / (kk == (real)0 ? (real)1 : kk);
    cx<real> d0 = du[e], d1 = du[e + count], d2 = du[e + 2 * count];
    const cx<real> u0 = u[e], u1 = u[e + count], u2 = u[e + 2 * count];
    const real px = d0.x * (kx * inv) + d1.x * (ky * inv) + d2.x * (kz * inv);
    const real py = d0.y * (kx * inv) + d1.y * (ky * inv) + d2.y * (kz * inv);
    const real v = nu * kk;
    d0.x -= px * kx;  d0.y -= py * kx;
    d1.x -= px * ky;  d1.y -= py * ky;
    d2.x -= px * kz;  d2.y -= py * kz;
    d0.x -= v * u0.x; d0.y -= v * u0.y;
    d1.x -= v * u1.x; d1.y -= v * u1.y;
    d2.x -= v * u2.x; d2.y -= v * u2.y;
    du[e] = d0;
    du[e + count] = d1;
    du[e + 2 * count] = d2;
  }
}

// Runge-Kutta stage on `count` real scalars: u = u0 + cb du (if u != null); u1 += ca du
template <typename real>
__global__ void __launch_bounds__(PS_THREADS)
ps_rk_kernel(real *__restrict__ u, const real *__restrict__ u0, real *__restrict__ u1, const real *__restrict__ du,
             int64_t count, real cb, real ca) {
  for (int64_t e = (int64_t)blockIdx.x * PS_THREADS + threadIdx.x; e < count; e += (int64_t)gridDim.x * PS_THREADS) {
    const real d = du[e];
    if (u) u[e] = u0[e] + cb * d;
    u1[e] += ca * d;
  }
}

}  // namespace

hipError_t launch_ps_curl(const void *u, void *out, const void *k0, const void *k1, const void *k2, int64_t n0,
                          int64_t n1, int64_t n2, int precision, hipStream_t s) {
  const int64_t count = n0 * n1 * n2;
  if (!count) return hipSuccess;
  if (precision == 8)
    hipLaunchKernelGGL(ps_curl_kernel<double>, dim3(ps_grid(count)), dim3(PS_THREADS), 0, s, (const cx<double> *)u,
                       (cx<double> *)out, (const double *)k0, (const double *)k1, (const double *)k2, n1, n2, count);
  else
    hipLaunchKernelGGL(ps_curl_kernel<float>, dim3(ps_grid(count)), dim3(PS_THREADS), 0, s, (const cx<float> *)u,
                       (cx<float> *)out, (const float *)k0, (const float *)k1, (const float *)k2, n1, n2, count);
  return hipGetLastError();
}

hipError_t launch_ps_cross(const void *a, const void *b, void *out, int64_t count, int precision, hipStream_t s) {
  if (!count) return hipSuccess;
  if (precision == 8)
    hipLaunchKernelGGL(ps_cross_kernel<double>, dim3(ps_grid(count)), dim3(PS_THREADS), 0, s, (const double *)a,
                       (const double *)b, (double *)out, count);
  else
    hipLaunchKernelGGL(ps_cross_kernel<float>, dim3(ps_grid(count)), dim3(PS_THREADS), 0, s, (const float *)a,
                       (const float *)b, (float *)out, count);
  return hipGetLastError();
}

hipError_t launch_ps_project(void *du, const void *u, const void *k0, const void *k1, const void *k2, int64_t n0,
                             int64_t n1, int64_t n2, double nu, int precision, hipStream_t s) {
  const int64_t count = n0 * n1 * n2;
  if (!count) return hipSuccess;
  if (precision == 8)
    hipLaunchKernelGGL(ps_project_kernel<double>, dim3(ps_grid(count)), dim3(PS_THREADS), 0, s, (cx<double> *)du,
                       (const cx<double> *)u, (const double *)k0, (const double *)k1, (const double *)k2, n1, n2,
                       count, nu);
  else
    hipLaunchKernelGGL(ps_project_kernel<float>, dim3(ps_grid(count)), dim3(PS_THREADS), 0, s, (cx<float> *)du,
                       (const cx<float> *)u, (const float *)k0, (const float *)k1, (const float *)k2, n1, n2, count,
                       (float)nu);
  return hipGetLastError();
}

hipError_t launch_ps_rk(void *u, const void *u0, void *u1, const void *du, int64_t count, double cb, double ca,
                        int precision, hipStream_t s) {
  if (!count) return hipSuccess;
  if (precision == 8)
    hipLaunchKernelGGL(ps_rk_kernel<double>, dim3(ps_grid(count)), dim3(PS_THREADS), 0, s, (double *)u,
                       (const double *)u0, (double *)u1, (const double *)du, count, cb, ca);
  else
    hipLaunchKernelGGL(ps_rk_kernel<float>, dim3(ps_grid(count)), dim3(PS_THREADS), 0, s, (float *)u, (const float *)u0,
                       (float *)u1, (const float *)du, count, (float)cb, (float)ca);
  return hipGetLastError();
}

}  // namespace gfft
