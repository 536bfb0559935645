// Generic any-length 1-D pass: LDS-resident mixed-radix Stockham (radices from a runtime factor
// list; 2/3/4 unrolled, any other prime by its O(r^2) definition).  This is the completeness
// path (odd sizes, primes, ragged tiles); power-of-two lengths take fft_pow2_impl.h instead.
//
// One workgroup = one tile of T columns; two ping-pong LDS buffers of T*n complex each.
// Loads/stores run with lanes along whichever of (element, column) is unit-stride in memory.
#include "gfft_internal.h"
#include "pass_io.h"

namespace gfft {

constexpr int GEN_THREADS = 256;

// radix-P butterfly for a small odd prime, operands and the P roots of unity in registers:
// P*P constant-index complex multiply-adds instead of the P*P table lookups of the fallback loop
template <typename real, int P>
__device__ __forceinline__ void prime_butterfly(const cx<real> *__restrict__ X, cx<real> *__restrict__ Y,
                                                int j, int j0, int k, int m, int Ns, int tstep, int rstep,
                                                const cx<real> *__restrict__ tw) {
  cx<real> x[P], w[P];
#pragma unroll
  for (int q = 0; q < P; ++q) {
    x[q] = X[j + q * m];
    if (q && k) x[q] = cmul(x[q], tw[q * k * tstep]);
    w[q] = tw[q * rstep];
  }
#pragma unroll
  for (int p = 0; p < P; ++p) {
    cx<real> acc = x[0];
#pragma unroll
    for (int q = 1; q < P; ++q) acc = acc + cmul(x[q], w[(p * q) % P]);
    Y[j0 + p * Ns] = acc;
  }
}

template <typename real>
__device__ __forceinline__ void generic_butterfly(const cx<real> *__restrict__ X, cx<real> *__restrict__ Y,
                                                  int j, int r, int Ns, int n,
                                                  const cx<real> *__restrict__ tw) {
  const int m = n / r;
  const int k = j % Ns;
  const int j0 = (j / Ns) * Ns * r + k;
  const int tstep = n / (Ns * r);  // twiddle-table stride of W_{Ns*r}
  if (r == 2) {
    cx<real> a = X[j], b = X[j + m];
    if (k) b = cmul(b, tw[k * tstep]);
    Y[j0] = a + b;
    Y[j0 + Ns] = a - b;
  } else if (r == 4) {
    cx<real> a0 = X[j], a1 = X[j + m], a2 = X[j + 2 * m], a3 = X[j + 3 * m];
    if (k) {
      a1 = cmul(a1, tw[k * tstep]);
      a2 = cmul(a2, tw[2 * k * tstep]);
      a3 = cmul(a3, tw[3 * k * tstep]);
    }
    cx<real> t0 = a0 + a2, t1 = a0 - a2, t2 = a1 + a3, t3 = mul_mi(a1 - a3);
    Y[j0] = t0 + t2;
    Y[j0 + Ns] = t1 + t3;
    Y[j0 + 2 * Ns] = t0 - t2;
    Y[j0 + 3 * Ns] = t1 - t3;
  } else if (r == 3) {
    cx<real> a = X[j], b = X[j + m], c = X[j + 2 * m];
    if (k) {
      b = cmul(b, tw[k * tstep]);
      c = cmul(c, tw[2 * k * tstep]);
    }
    const real h = (real)0.86602540378443864676372317075294;  // sqrt(3)/2
    cx<real> t1 = b + c;
    cx<real> t2 = {a.x - (real)0.5 * t1.x, a.y - (real)0.5 * t1.y};
    cx<real> t3 = {(b.x - c.x) * h, (b.y - c.y) * h};
    Y[j0] = a + t1;
    Y[j0 + Ns] = {t2.x + t3.y, t2.y - t3.x};
    Y[j0 + 2 * Ns] = {t2.x - t3.y, t2.y + t3.x};
  } else if (r == 5) {
    prime_butterfly<real, 5>(X, Y, j, j0, k, m, Ns, tstep, n / 5, tw);
  } else if (r == 7) {
    prime_butterfly<real, 7>(X, Y, j, j0, k, m, Ns, tstep, n / 7, tw);
  } else if (r == 11) {
    prime_butterfly<real, 11>(X, Y, j, j0, k, m, Ns, tstep, n / 11, tw);
  } else if (r == 13) {
    prime_butterfly<real, 13>(X, Y, j, j0, k, m, Ns, tstep, n / 13, tw);
  } else {
    // y[p] = sum_q x[q] * W_{Ns r}^{q k} * W_r^{p q}
    const int rstep = n / r;
    for (int p = 0; p < r; ++p) {
      cx<real> acc = {0, 0};
      int pq = 0;  // (p*q) mod r
      for (int q = 0; q < r; ++q) {
        cx<real> x = X[j + q * m];
        int ti = q * k * tstep + pq * rstep;  // < 2n
        if (ti >= n) ti -= n;
        x = cmul(x, tw[ti]);
        acc = acc + x;
        pq += p;
        if (pq >= r) pq -= r;
      }
      Y[j0 + p * Ns] = acc;
    }
  }
}

template <typename real>
__global__ void __launch_bounds__(GEN_THREADS)
fft_generic_kernel(PassDesc d, Factors f, int T, const void *__restrict__ in, void *__restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int n = d.n;
  cx<real> *X = reinterpret_cast<cx<real> *>(smem);
  cx<real> *Y = X + (size_t)T * n;
  int64_t *colbase = reinterpret_cast<int64_t *>(Y + (size_t)T * n);  // [3][T]: in, out, mid
  const cx<real> *tw = reinterpret_cast<const cx<real> *>(d.tw);
  const int tid = threadIdx.x;

  for (int64_t tile = blockIdx.x; tile * T < d.batch; tile += gridDim.x) {
    const int64_t b0 = tile * T;
    const int ncols = (int)((d.batch - b0) < (int64_t)T ? (d.batch - b0) : (int64_t)T);
    __syncthreads();
    for (int c = tid; c < ncols; c += GEN_THREADS) {
      ColAddr a = column_address(d, b0 + c);
      colbase[c] = a.in;
      colbase[T + c] = a.out;
      colbase[2 * T + c] = a.mid;
    }
    __syncthreads();
    // ---- load
    const int nin = n;  // every logical element is materialised (r2c: imag 0; c2r: Hermitian mirror)
    const int total = ncols * nin;
    if (d.in_es == 1 || ncols == 1) {
      for (int idx = tid; idx < total; idx += GEN_THREADS) {
        int c = idx / nin, e = idx - c * nin;
        X[c * n + e] = load_elem<real>(d, in, colbase[c], e);
      }
    } else {
      for (int idx = tid; idx < total; idx += GEN_THREADS) {
        int e = idx / ncols, c = idx - e * ncols;
        X[c * n + e] = load_elem<real>(d, in, colbase[c], e);
      }
    }
    __syncthreads();
    // ---- Stockham stages
    cx<real> *A = X, *B = Y;
    int Ns = 1;
    for (int s = 0; s < f.count; ++s) {
      const int r = f.r[s];
      const int m = n / r;
      const int work = ncols * m;
      for (int idx = tid; idx < work; idx += GEN_THREADS) {
        int c = idx / m, j = idx - c * m;
        generic_butterfly<real>(A + c * n, B + c * n, j, r, Ns, n, tw);
      }
      __syncthreads();
      cx<real> *t = A;
      A = B;
      B = t;
      Ns *= r;
    }
    // ---- store
    const int nout = (d.mode == MODE_R2C) ? n / 2 + 1 : n;
    const int stotal = ncols * nout;
    if (d.out_es == 1 || ncols == 1) {
      for (int idx = tid; idx < stotal; idx += GEN_THREADS) {
        int c = idx / nout, e = idx - c * nout;
        store_elem<real>(d, out, colbase[T + c], e, colbase[2 * T + c], A[c * n + e]);
      }
    } else {
      for (int idx = tid; idx < stotal; idx += GEN_THREADS) {
        int e = idx / ncols, c = idx - e * ncols;
        store_elem<real>(d, out, colbase[T + c], e, colbase[2 * T + c], A[c * n + e]);
      }
    }
  }
}

int generic_max_n(int precision) { return precision == 8 ? 4096 : 8192; }

// ---- lengths 3 / 5 / 7 / 11 / 13: one column per thread, everything in registers ----------------
// The small factor of a two-pass split (896 = 128 x 7, 1408 = 128 x 11, ...) has thousands of
// adjacent columns: lanes run along them (fully coalesced 16-byte accesses), each thread loads
// its A strided elements, evaluates the DFT by definition with the A roots of unity in registers
// (A*A constant-index multiply-adds), and stores.  No LDS, no synchronisation.
constexpr int TINY_THREADS = 256;

template <typename real, int A>
__global__ void __launch_bounds__(TINY_THREADS)
tiny_dft_kernel(PassDesc d, const void *__restrict__ in, void *__restrict__ out) {
  const cx<real> *tw = reinterpret_cast<const cx<real> *>(d.tw);
  cx<real> w[A];
#pragma unroll
  for (int q = 0; q < A; ++q) w[q] = tw[q];
  const real sy_in = d.conj_in ? (real)-1 : (real)1;
  const real sx = (real)d.scale, sy = d.conj_out ? -sx : sx;
  const cx<real> *src = reinterpret_cast<const cx<real> *>(in);
  cx<real> *dst = reinterpret_cast<cx<real> *>(out);
  for (int64_t b = (int64_t)blockIdx.x * TINY_THREADS + threadIdx.x; b < d.batch;
       b += (int64_t)gridDim.x * TINY_THREADS) {
    const int64_t bm = b / d.inner, i = b - bm * d.inner, o = bm / d.mid, m = bm - o * d.mid;
    const int64_t in0 = o * d.in_os + m * d.in_ms + i * d.in_is;
    const int64_t out0 = o * d.out_os + m * d.out_ms + i * d.out_is;
    cx<real> x[A];
#pragma unroll
    for (int q = 0; q < A; ++q) {
      x[q] = src[in0 + q * d.in_es];
      x[q].y *= sy_in;
    }
#pragma unroll
    for (int p = 0; p < A; ++p) {
      cx<real> acc = x[0];
#pragma unroll
      for (int q = 1; q < A; ++q) acc = acc + cmul(x[q], w[(p * q) % A]);
      dst[out0 + p * d.out_es] = {acc.x * sx, acc.y * sy};
    }
  }
}

template <typename real, int A>
static hipError_t launch_tiny(const PassDesc &d, const void *in, void *out, hipStream_t s) {
  const int64_t blocks = (d.batch + TINY_THREADS - 1) / TINY_THREADS;
  const int grid = (int)(blocks < 8192 ? blocks : 8192);
  hipLaunchKernelGGL((tiny_dft_kernel<real, A>), dim3(grid), dim3(TINY_THREADS), 0, s, d, in, out);
  return hipGetLastError();
}

template <typename real>
static bool try_tiny(const PassDesc &d, const void *in, void *out, hipStream_t s, hipError_t *err) {
  // worthwhile when lanes can run along adjacent columns
  if (d.mode != MODE_C2C || d.tw_hi || d.tr_dir || d.in_is != 1 || d.out_is != 1 || d.inner < 64) return false;
  switch (d.n) {
    case 3: *err = launch_tiny<real, 3>(d, in, out, s); return true;
    case 5: *err = launch_tiny<real, 5>(d, in, out, s); return true;
    case 7: *err = launch_tiny<real, 7>(d, in, out, s); return true;
    case 11: *err = launch_tiny<real, 11>(d, in, out, s); return true;
    case 13: *err = launch_tiny<real, 13>(d, in, out, s); return true;
  }
  return false;
}

template <typename real>
static hipError_t launch_generic_t(const PassDesc &d, const Factors &f, const void *in, void *out,
                                   hipStream_t s) {
  hipError_t terr;
  if (try_tiny<real>(d, in, out, s, &terr)) return terr;
  const size_t esz = sizeof(cx<real>);
  // columns per tile: aim for ~32 KiB per ping-pong buffer, at least 1, at most 64 (wider tiles measured slower)
  int T = (int)((32 * 1024) / ((size_t)d.n * esz));
  if (T < 1) T = 1;
  if (T > 64) T = 64;
  if ((int64_t)T > d.batch) T = (int)d.batch;
  // keep the chip busy: prefer >= 1024 tiles when the batch allows
  while (T > 8 && (d.batch + T - 1) / T < 1024) T /= 2;
  size_t lds = 2 * (size_t)T * d.n * esz + 3 * (size_t)T * sizeof(int64_t);
  static bool attr_set[kMaxDevices] = {};       // (per device)
  const int dev = current_device();
  if (!attr_set[dev]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&fft_generic_kernel<real>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    attr_set[dev] = true;
  }
  int64_t tiles = (d.batch + T - 1) / T;
  int grid = (int)(tiles < 4096 ? tiles : 4096);
  hipLaunchKernelGGL(fft_generic_kernel<real>, dim3(grid), dim3(GEN_THREADS), lds, s, d, f, T, in, out);
  return hipGetLastError();
}

hipError_t launch_generic(const PassDesc &d, const Factors &f, int precision, const void *in,
                          void *out, hipStream_t s) {
  if (d.batch <= 0) return hipSuccess;
  if (precision == 8) return launch_generic_t<double>(d, f, in, out, s);
  return launch_generic_t<float>(d, f, in, out, s);
}

}  // namespace gfft
