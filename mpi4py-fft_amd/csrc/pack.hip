// Data-movement kernels around the transforms:
//   pack / unpack   what MPI's subarray datatypes do inside Alltoallw (pencil.py:12-29,182,200)
//   truncate / pad  3/2-rule dealiasing copies (libfft.py:263-311) with the scale fused
//   scale           `array *= M` (libfft.py:412-413)
//   copy probes     HBM ceilings quoted next to the roofline numbers
// All are pure streaming kernels: 16-byte accesses where the geometry allows, grid-stride loops.
#include "gfft_internal.h"

namespace gfft {

constexpr int MV_THREADS = 256;
constexpr int MV_MAX_BLOCKS = 256 * 16;

static inline int mv_grid(int64_t work) {
  int64_t b = (work + MV_THREADS - 1) / MV_THREADS;
  if (b > MV_MAX_BLOCKS) b = MV_MAX_BLOCKS;
  if (b < 1) b = 1;
  return (int)b;
}

// block p of an axis of length N cut into P blocks (pencil.py:5-9): first r blocks have q+1
struct BlockRule {
  int64_t q, r;
  __host__ __device__ void locate(int64_t a, int64_t &p, int64_t &start, int64_t &len) const {
    const int64_t big = r * (q + 1);
    if (a < big) {
      p = a / (q + 1);
      start = p * (q + 1);
      len = q + 1;
    } else {
      p = r + (a - big) / q;
      start = big + (p - r) * q;
      len = q;
    }
  }
};

// array [outer][naxis][inner] (units of U bytes) <-> blocks laid out one after another, block p
// being the row-major [outer][len_p][inner] sub-array starting at outer*inner*start_p units.
template <typename U, bool UNPACK>
__global__ void __launch_bounds__(MV_THREADS)
pack_kernel(const U *__restrict__ src, U *__restrict__ dst, int64_t outer, int64_t naxis,
            int64_t inner, BlockRule rule) {
  const int64_t total = outer * naxis * inner;
  const int64_t row = naxis * inner;
  for (int64_t idx = (int64_t)blockIdx.x * MV_THREADS + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * MV_THREADS) {
    const int64_t o = idx / row;
    const int64_t rem = idx - o * row;
    const int64_t a = rem / inner;
    const int64_t i = rem - a * inner;
    int64_t p, start, len;
    rule.locate(a, p, start, len);
    const int64_t packed = outer * inner * start + (o * len + (a - start)) * inner + i;
    if (UNPACK) dst[idx] = src[packed];
    else dst[packed] = src[idx];
  }
}

struct alignas(16) u128 { unsigned long long a, b; };

hipError_t launch_pack(const void *src, void *dst, int64_t outer, int64_t naxis, int64_t inner,
                       int nparts, int itemsize, bool unpack, hipStream_t s) {
  if (outer * naxis * inner == 0) return hipSuccess;
  BlockRule rule{naxis / nparts, naxis % nparts};
  const int64_t rowbytes = inner * itemsize;
  const bool al16 = (rowbytes % 16 == 0) && (((uintptr_t)src | (uintptr_t)dst) % 16 == 0);
#define PK(U, UNP, INNER) \
  hipLaunchKernelGGL((pack_kernel<U, UNP>), dim3(mv_grid(outer * naxis * (INNER))), dim3(MV_THREADS), 0, s, \
                     (const U *)src, (U *)dst, outer, naxis, (INNER), rule)
  if (al16) {
    if (unpack) PK(u128, true, rowbytes / 16); else PK(u128, false, rowbytes / 16);
  } else if (itemsize == 16) {
    if (unpack) PK(u128, true, inner); else PK(u128, false, inner);
  } else if (itemsize == 8) {
    if (unpack) PK(unsigned long long, true, inner); else PK(unsigned long long, false, inner);
  } else if (itemsize == 4) {
    if (unpack) PK(unsigned int, true, inner); else PK(unsigned int, false, inner);
  } else {
    return hipErrorInvalidValue;
  }
#undef PK
  return hipGetLastError();
}

// ---- truncation / padding ---------------------------------------------------------------------
// PAD == false: trunc[o][k][i] = scale * f(padded)   (libfft.py:263-284)
// PAD == true : padded[o][kp][i] = g(trunc)          (libfft.py:286-311)
template <typename real, bool PAD>
__global__ void __launch_bounds__(MV_THREADS)
trunc_kernel(const cx<real> *__restrict__ src, cx<real> *__restrict__ dst, int64_t outer,
             int64_t npad, int64_t N, int64_t inner, int is_real, real scale) {
  const int64_t nd = PAD ? npad : N;        // destination axis length
  const int64_t ns = PAD ? N : npad;        // source axis length
  const int64_t total = outer * nd * inner;
  const int64_t h = N / 2;
  const bool even = (N % 2) == 0;           // parity rule is on the TRUNCATED length (libfft.py:267,290)
  for (int64_t idx = (int64_t)blockIdx.x * MV_THREADS + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * MV_THREADS) {
    const int64_t o = idx / (nd * inner);
    const int64_t rem = idx - o * nd * inner;
    const int64_t k = rem / inner;
    const int64_t i = rem - k * inner;
    const cx<real> *s = src + o * ns * inner + i;
    cx<real> v = {0, 0};
    if (!PAD) {
      if (is_real) {
        v = s[k * inner];
        if (even && k == N - 1) { v.x *= 2; v.y = 0; }
      } else {
        if (k <= h) v = s[k * inner];
        if (h > 0 && k >= N - h) v = v + s[(npad - N + k) * inner];
      }
      v.x *= scale;
      v.y *= scale;
    } else {
      if (is_real) {
        if (k < N) {
          v = s[k * inner];
          if (even && k == N - 1) { v.x *= (real)0.5; v.y = 0; }
        }
      } else {
        if (k <= h) v = s[k * inner];
        else if (h > 0 && k >= npad - h) v = s[(N - npad + k) * inner];
        if (even && (k == h || k == npad - h)) { v.x *= (real)0.5; v.y *= (real)0.5; }
      }
    }
    dst[idx] = v;
  }
}

hipError_t launch_trunc(const void *src, void *dst, int64_t outer, int64_t npad,
                        int64_t ntrunc, int64_t inner, int is_real, int precision, double scale,
                        bool pad_direction, hipStream_t s) {
  const int64_t total = outer * (pad_direction ? npad : ntrunc) * inner;
  if (total == 0) return hipSuccess;
  const int grid = mv_grid(total);
  if (precision == 8) {
    if (pad_direction)
      hipLaunchKernelGGL((trunc_kernel<double, true>), dim3(grid), dim3(MV_THREADS), 0, s,
                         (const cx<double> *)src, (cx<double> *)dst, outer, npad, ntrunc, inner, is_real, 1.0);
    else
      hipLaunchKernelGGL((trunc_kernel<double, false>), dim3(grid), dim3(MV_THREADS), 0, s,
                         (const cx<double> *)src, (cx<double> *)dst, outer, npad, ntrunc, inner, is_real, scale);
  } else {
    if (pad_direction)
      hipLaunchKernelGGL((trunc_kernel<float, true>), dim3(grid), dim3(MV_THREADS), 0, s,
                         (const cx<float> *)src, (cx<float> *)dst, outer, npad, ntrunc, inner, is_real, 1.0f);
    else
      hipLaunchKernelGGL((trunc_kernel<float, false>), dim3(grid), dim3(MV_THREADS), 0, s,
                         (const cx<float> *)src, (cx<float> *)dst, outer, npad, ntrunc, inner, is_real, (float)scale);
  }
  return hipGetLastError();
}

// ---- embedding fallbacks (Bluestein / long real transforms) -----------------------------------
template <typename real>
__global__ void __launch_bounds__(MV_THREADS)
embed_kernel(PointDesc p, const void *__restrict__ in, cx<real> *__restrict__ W) {
  const int64_t total = p.outer * p.Lw * p.inner;
  const cx<real> *chirp = reinterpret_cast<const cx<real> *>(p.chirp);
  for (int64_t idx = (int64_t)blockIdx.x * MV_THREADS + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * MV_THREADS) {
    const int64_t o = idx / (p.Lw * p.inner);
    const int64_t rem = idx - o * p.Lw * p.inner;
    const int64_t j = rem / p.inner, i = rem - j * p.inner;
    cx<real> v = {0, 0};
    if (p.mode == MODE_R2R) {
      const int64_t jj = j - p.pos0;
      if (jj >= 0 && jj < p.n) {
        const real x = reinterpret_cast<const real *>(in)[(o * p.nin + jj) * p.inner + i];
        const cx<real> a = reinterpret_cast<const cx<real> *>(p.pre)[jj];
        v = {a.x * x, a.y * x};
      }
    } else if (j < p.n) {
      if (p.mode == MODE_R2C) {
        v.x = reinterpret_cast<const real *>(in)[(o * p.nin + j) * p.inner + i];
      } else if (p.mode == MODE_C2R) {
        const bool mirror = j > p.n / 2;
        const int64_t jj = mirror ? p.n - j : j;
        v = reinterpret_cast<const cx<real> *>(in)[(o * p.nin + jj) * p.inner + i];
        if (mirror) v.y = -v.y;
      } else {
        v = reinterpret_cast<const cx<real> *>(in)[(o * p.nin + j) * p.inner + i];
      }
      if (p.conj) v.y = -v.y;
      if (chirp) v = cmul(v, chirp[j]);
    }
    W[idx] = v;
  }
}

template <typename real>
__global__ void __launch_bounds__(MV_THREADS) mulb_kernel(PointDesc p, cx<real> *__restrict__ W) {
  const int64_t total = p.outer * p.Lw * p.inner;
  const cx<real> *B = reinterpret_cast<const cx<real> *>(p.B);
  for (int64_t idx = (int64_t)blockIdx.x * MV_THREADS + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * MV_THREADS) {
    const int64_t k = (idx / p.inner) % p.Lw;
    W[idx] = cmul(W[idx], B[k]);
  }
}

template <typename real>
__global__ void __launch_bounds__(MV_THREADS)
extract_kernel(PointDesc p, const cx<real> *__restrict__ W, void *__restrict__ out, real scale) {
  const int64_t total = p.outer * p.nout * p.inner;
  const cx<real> *chirp = reinterpret_cast<const cx<real> *>(p.chirp);
  for (int64_t idx = (int64_t)blockIdx.x * MV_THREADS + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * MV_THREADS) {
    const int64_t o = idx / (p.nout * p.inner);
    const int64_t rem = idx - o * p.nout * p.inner;
    const int64_t k = rem / p.inner, i = rem - k * p.inner;
    if (p.mode == MODE_R2R) {
      const cx<real> z = W[(o * p.Lw + k + p.idx0) * p.inner + i];
      const cx<real> b = reinterpret_cast<const cx<real> *>(p.post)[k];
      reinterpret_cast<real *>(out)[idx] = (b.x * z.x - b.y * z.y) * scale;
      continue;
    }
    cx<real> v = W[(o * p.Lw + k) * p.inner + i];
    if (chirp) v = cmul(v, chirp[k]);
    if (p.conj) v.y = -v.y;
    v.x *= scale;
    v.y *= scale;
    if (p.mode == MODE_C2R) reinterpret_cast<real *>(out)[idx] = v.x;
    else reinterpret_cast<cx<real> *>(out)[idx] = v;
  }
}

hipError_t launch_embed(const PointDesc &p, int precision, const void *in, void *scratch, hipStream_t s) {
  const int64_t total = p.outer * p.Lw * p.inner;
  if (total == 0) return hipSuccess;
  if (precision == 8)
    hipLaunchKernelGGL(embed_kernel<double>, dim3(mv_grid(total)), dim3(MV_THREADS), 0, s, p, in, (cx<double> *)scratch);
  else
    hipLaunchKernelGGL(embed_kernel<float>, dim3(mv_grid(total)), dim3(MV_THREADS), 0, s, p, in, (cx<float> *)scratch);
  return hipGetLastError();
}

hipError_t launch_mulb(const PointDesc &p, int precision, void *scratch, hipStream_t s) {
  const int64_t total = p.outer * p.Lw * p.inner;
  if (total == 0) return hipSuccess;
  if (precision == 8)
    hipLaunchKernelGGL(mulb_kernel<double>, dim3(mv_grid(total)), dim3(MV_THREADS), 0, s, p, (cx<double> *)scratch);
  else
    hipLaunchKernelGGL(mulb_kernel<float>, dim3(mv_grid(total)), dim3(MV_THREADS), 0, s, p, (cx<float> *)scratch);
  return hipGetLastError();
}

hipError_t launch_extract(const PointDesc &p, int precision, const void *scratch, void *out, double scale, hipStream_t s) {
  const int64_t total = p.outer * p.nout * p.inner;
  if (total == 0) return hipSuccess;
  if (precision == 8)
    hipLaunchKernelGGL(extract_kernel<double>, dim3(mv_grid(total)), dim3(MV_THREADS), 0, s, p, (const cx<double> *)scratch, out, scale);
  else
    hipLaunchKernelGGL(extract_kernel<float>, dim3(mv_grid(total)), dim3(MV_THREADS), 0, s, p, (const cx<float> *)scratch, out, (float)scale);
  return hipGetLastError();
}

// ---- scale ----------------------------------------------------------------------------------
template <typename real>
__global__ void __launch_bounds__(MV_THREADS) scale_kernel(real *__restrict__ p, int64_t n, real sc) {
  for (int64_t i = (int64_t)blockIdx.x * MV_THREADS + threadIdx.x; i < n; i += (int64_t)gridDim.x * MV_THREADS)
    p[i] *= sc;
}

hipError_t launch_scale(void *data, int64_t count, int precision, double scale, hipStream_t s) {
  if (count == 0) return hipSuccess;
  if (precision == 8)
    hipLaunchKernelGGL(scale_kernel<double>, dim3(mv_grid(count)), dim3(MV_THREADS), 0, s, (double *)data, count, scale);
  else
    hipLaunchKernelGGL(scale_kernel<float>, dim3(mv_grid(count)), dim3(MV_THREADS), 0, s, (float *)data, count, (float)scale);
  return hipGetLastError();
}

// ---- probes ---------------------------------------------------------------------------------
typedef unsigned int u4 __attribute__((ext_vector_type(4)));

template <bool NT>
__global__ void __launch_bounds__(MV_THREADS) copy_kernel(const u4 *__restrict__ src, u4 *__restrict__ dst, int64_t n) {
  // 4 x 16 B per thread in flight, each a fully coalesced 4 KiB per workgroup
  const int64_t chunk = 4 * MV_THREADS;
  for (int64_t base = (int64_t)blockIdx.x * chunk; base < n; base += (int64_t)gridDim.x * chunk) {
    u4 v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int64_t i = base + k * MV_THREADS + threadIdx.x;
      if (i < n) v[k] = NT ? __builtin_nontemporal_load(src + i) : src[i];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int64_t i = base + k * MV_THREADS + threadIdx.x;
      if (i < n) {
        if (NT) __builtin_nontemporal_store(v[k], dst + i);
        else dst[i] = v[k];
      }
    }
  }
}

// The streaming ceiling bench.py quotes.  Round 4 swept the copy over accesses in flight, cache policy, workgroup
// size and count, tile order and size on a 16 GiB pair (tools/probes/copy_sweep.hip, profiles/r04_copy_sweep.txt):
// the round-1 geometry above (256 threads, 4 x 16 B in flight, plain policy) reached 5.1 TB/s on a box where
// 1024-thread workgroups with 8 x 16 B in flight per thread and non-temporal loads AND stores reached 5.9 TB/s
// (threads 256 -> 1024: +5 %, nt stores +3 %, nt loads +2 %, 8-16 workgroups' worth of chunks per CU; one
// contiguous share per workgroup instead of grid-stride chunks: -2 %).  Read-only streams reach 7.3 TB/s,
// write-only 6.3 TB/s; the same copy between freshly allocated buffers 5.4 - 6.5 TB/s by physical placement.
constexpr int CW_THREADS = 1024, CW_U = 8;
__global__ void __launch_bounds__(CW_THREADS) copy_wide_kernel(const u4 *__restrict__ src, u4 *__restrict__ dst, int64_t n) {
  const int64_t chunk = (int64_t)CW_U * CW_THREADS;
  for (int64_t base = (int64_t)blockIdx.x * chunk; base < n; base += (int64_t)gridDim.x * chunk) {
    u4 v[CW_U];
    if (base + chunk <= n) {
#pragma unroll
      for (int k = 0; k < CW_U; ++k) v[k] = __builtin_nontemporal_load(src + base + k * CW_THREADS + threadIdx.x);
#pragma unroll
      for (int k = 0; k < CW_U; ++k) __builtin_nontemporal_store(v[k], dst + base + k * CW_THREADS + threadIdx.x);
    } else {
      for (int k = 0; k < CW_U; ++k) {
        const int64_t i = base + k * CW_THREADS + threadIdx.x;
        if (i < n) dst[i] = src[i];
      }
    }
  }
}

int g_copy_nt = 2;      // option "copy_nt": 2 = the tuned geometry above; 0 / 1 = the round-1 geometry, plain / non-temporal

hipError_t launch_copy(const void *src, void *dst, size_t bytes, hipStream_t s) {
  const int64_t n = (int64_t)(bytes / 16);
  if (n == 0) return hipSuccess;
  if (g_copy_nt == 2) {
    int64_t blocks = (n + CW_U * CW_THREADS - 1) / (CW_U * CW_THREADS);
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(copy_wide_kernel, dim3((int)blocks), dim3(CW_THREADS), 0, s, (const u4 *)src, (u4 *)dst, n);
    return hipGetLastError();
  }
  int64_t blocks = (n + 4 * MV_THREADS - 1) / (4 * MV_THREADS);
  if (blocks > 256 * 32) blocks = 256 * 32;
  if (g_copy_nt)
    hipLaunchKernelGGL(copy_kernel<true>, dim3((int)blocks), dim3(MV_THREADS), 0, s, (const u4 *)src, (u4 *)dst, n);
  else
    hipLaunchKernelGGL(copy_kernel<false>, dim3((int)blocks), dim3(MV_THREADS), 0, s, (const u4 *)src, (u4 *)dst, n);
  return hipGetLastError();
}

// column-pass access pattern without arithmetic: a workgroup moves a [n][tcols] tile of 16-byte
// elements (rows `inner` elements apart), 8 rows in flight per thread.
__global__ void __launch_bounds__(MV_THREADS)
tile_copy_kernel(const u128 *__restrict__ src, u128 *__restrict__ dst, int64_t outer, int64_t n,
                 int64_t inner, int tcols) {
  const int64_t tiles_per_outer = inner / tcols;
  const int64_t ntiles = outer * tiles_per_outer;
  const int c = threadIdx.x % tcols;
  const int r0 = threadIdx.x / tcols;
  const int rstep = MV_THREADS / tcols;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t o = tile / tiles_per_outer;
    const int64_t i0 = (tile - o * tiles_per_outer) * tcols;
    const int64_t base = o * n * inner + i0 + c;
    for (int64_t e = r0; e < n; e += 8 * rstep) {
      u128 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (e + u * rstep < n) v[u] = src[base + (e + u * rstep) * inner];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (e + u * rstep < n) dst[base + (e + u * rstep) * inner] = v[u];
    }
  }
}

hipError_t launch_tile_copy(const void *src, void *dst, int64_t outer, int64_t n, int64_t inner,
                            int tcols, hipStream_t s) {
  if (tcols < 1 || tcols > MV_THREADS || (MV_THREADS % tcols) || (inner % tcols)) return hipErrorInvalidValue;
  const int64_t ntiles = outer * (inner / tcols);
  const int grid = (int)(ntiles < MV_MAX_BLOCKS ? ntiles : MV_MAX_BLOCKS);
  hipLaunchKernelGGL(tile_copy_kernel, dim3(grid), dim3(MV_THREADS), 0, s, (const u128 *)src, (u128 *)dst, outer, n, inner, tcols);
  return hipGetLastError();
}

}  // namespace gfft
