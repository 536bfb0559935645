// Developer probe: does the 256 MiB Infinity Cache carry a producer -> consumer hand-off between two
// passes over a large array when both passes run inside ONE persistent kernel (no launch tails)?
//
//   baseline : B = f(A) over the whole array, then C = g(B) over the whole array (two launches)
//   fused    : one launch; workgroups draw tickets; a ticket is (phase, plane, tile):
//                phase 1 copies a tile of plane p from A into slot p % RING of a small ring buffer,
//                phase 2 copies a tile of plane p from the ring into C -- reading "columns": a phase-2
//                tile touches every phase-1 tile of its plane, as the second pass of a 2-D transform does,
//              so phase 2 of plane p waits (acquire) until all phase-1 tiles of p are published (release),
//              and phase 1 of plane p + RING waits until all phase-2 tiles of plane p are done.
//              Tickets are ordered so that a ticket only ever waits for lower tickets: no deadlock
//              whatever the number of resident workgroups.
// Build: hipcc -O3 --offload-arch=gfx950 mall_ring_probe.hip -o mall_ring_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

typedef float vec4 __attribute__((ext_vector_type(4)));

constexpr int THREADS = 1024;
constexpr int ROWS = 1024;             // a plane is ROWS x ROWS 16-byte elements = 16 MiB
constexpr int TILE = 16;               // rows (phase 1) / columns (phase 2) per tile -> 64 tiles per plane-phase
constexpr int TPP = ROWS / TILE;

// 16-byte accesses at SYSTEM scope (sc0 sc1): the store is written through the XCD's L2 to the memory side
// (Infinity Cache / HBM), the load never hits a (possibly stale) L2 line -- coherence between workgroups on
// different XCDs per access, instead of writing back / invalidating whole L2s at every hand-off
typedef unsigned uvec4 __attribute__((ext_vector_type(4)));
// (raw buffer accesses with cache policy aux = sc0 | sc1 = 1 | 16: the compiler tracks their completion itself)
struct SysBuf {
  __amdgpu_buffer_rsrc_t r;
  __device__ __forceinline__ explicit SysBuf(const void *base) : r(__builtin_amdgcn_make_buffer_rsrc((void *)base, 0, 0x7fffffff, 0x00020000)) {}
  __device__ __forceinline__ void st(unsigned elem, vec4 v) const {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uvec4, v), r, elem * 16u, 0, 17);
  }
  __device__ __forceinline__ vec4 ld(unsigned elem) const {
    return __builtin_bit_cast(vec4, __builtin_amdgcn_raw_buffer_load_b128(r, elem * 16u, 0, 17));
  }
};
template <bool SYS_ST>
__device__ __forceinline__ void copy_rows_t(const vec4 *__restrict__ src, vec4 *__restrict__ dst, int t) {
  const size_t base = (size_t)t * TILE * ROWS;
  const SysBuf sb(dst);                 // dst = the plane's ring slot (16 MiB: 32-bit offsets)
#pragma unroll
  for (int q = 0; q < TILE; ++q) {
    const vec4 v = src[base + (size_t)q * ROWS + threadIdx.x] * 1.0000001f;
    if (SYS_ST) sb.st((unsigned)(base + (size_t)q * ROWS + threadIdx.x), v); else dst[base + (size_t)q * ROWS + threadIdx.x] = v;
  }
  if (SYS_ST) __builtin_amdgcn_s_waitcnt(0);       // every write-through store acknowledged before the flag goes up
}
template <bool SYS_LD>
__device__ __forceinline__ void copy_cols_t(const vec4 *__restrict__ src, vec4 *__restrict__ dst, int t) {
  const int c = threadIdx.x % TILE, r0 = threadIdx.x / TILE;
  constexpr int Q = ROWS / (THREADS / TILE);
  const SysBuf sb(src);
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    const size_t i = (size_t)(r0 + q * (THREADS / TILE)) * ROWS + (size_t)t * TILE + c;
    const vec4 v = SYS_LD ? sb.ld((unsigned)i) : src[i];
    dst[i] = v * 1.0000001f;
  }
}
__global__ void diff_kernel(const vec4 *a, const vec4 *b, size_t n, unsigned *bad) {
  unsigned cnt = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const vec4 x = a[i], y = b[i];
    cnt += (x.x != y.x) | (x.y != y.y) | (x.z != y.z) | (x.w != y.w);
  }
  if (cnt) atomicAdd(bad, cnt);
}

// phase 1: rows [t*16, t*16+16) of the plane, contiguous: 16 rows x 16 KiB
__device__ __forceinline__ void copy_rows(const vec4 *__restrict__ src, vec4 *__restrict__ dst, int t) {
  const size_t base = (size_t)t * TILE * ROWS;
#pragma unroll
  for (int q = 0; q < TILE; ++q) {
    const size_t i = base + (size_t)q * ROWS + threadIdx.x;
    dst[i] = src[i] * 1.0000001f;
  }
}
// phase 2: columns [t*16, t*16+16) of the plane: 1024 segments of 256 bytes, stride 16 KiB
__device__ __forceinline__ void copy_cols(const vec4 *__restrict__ src, vec4 *__restrict__ dst, int t) {
  const int c = threadIdx.x % TILE, r0 = threadIdx.x / TILE;       // 64 rows per sweep
#pragma unroll
  for (int q = 0; q < ROWS / (THREADS / TILE); ++q) {
    const size_t i = (size_t)(r0 + q * (THREADS / TILE)) * ROWS + (size_t)t * TILE + c;
    dst[i] = src[i] * 1.0000001f;
  }
}

__global__ void __launch_bounds__(THREADS) whole_rows(const vec4 *a, vec4 *b, int planes) {
  for (int k = blockIdx.x; k < planes * TPP; k += gridDim.x)
    copy_rows(a + (size_t)(k / TPP) * ROWS * ROWS, b + (size_t)(k / TPP) * ROWS * ROWS, k % TPP);
}
__global__ void __launch_bounds__(THREADS) whole_cols(const vec4 *a, vec4 *b, int planes) {
  for (int k = blockIdx.x; k < planes * TPP; k += gridDim.x)
    copy_cols(a + (size_t)(k / TPP) * ROWS * ROWS, b + (size_t)(k / TPP) * ROWS * ROWS, k % TPP);
}

__global__ void __launch_bounds__(THREADS)
fused(const vec4 *a, vec4 *ring, vec4 *c, unsigned *ctr, int planes, int ring_planes, int lag, int fence_mode) {
  // ctr[0] = ticket; ctr[16 + p] = phase-1 tiles of plane p published; ctr[16 + planes + p] = phase-2 tiles done
  unsigned *done1 = ctr + 16, *done2 = ctr + 16 + planes;
  __shared__ unsigned tk;
  const unsigned total = 2u * planes * TPP;
  for (;;) {
    if (threadIdx.x == 0) tk = atomicAdd(&ctr[0], 1u);
    __syncthreads();
    const unsigned k = __builtin_amdgcn_readfirstlane(tk);     // (uniform: scalar branches below)
    __syncthreads();
    if (k >= total) break;
    if (fence_mode & 4) continue;                              // debug: tickets only
    const unsigned s = k / TPP, t = k % TPP;
    int phase, p;
    if ((int)s < lag) { phase = 1; p = s; }
    else {
      const unsigned s2 = s - lag;
      const unsigned pairs = planes - lag;            // planes whose phase 2 is followed by a phase 1
      if (s2 < 2 * pairs) { p = s2 / 2; phase = (s2 & 1) ? 1 : 2; if (phase == 1) p += lag; }
      else { phase = 2; p = pairs + (s2 - 2 * pairs); }
    }
    const size_t plane = (size_t)ROWS * ROWS;
    vec4 *slot = ring + (size_t)(p % ring_planes) * plane;
    if (phase == 1) {
      if (p >= ring_planes && !(fence_mode & 8)) {
        if (threadIdx.x == 0) {
          unsigned spins = 0;
          while (__hip_atomic_load(&done2[p - ring_planes], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < TPP) {
            __builtin_amdgcn_s_sleep(8);
            if (++spins > (1u << 15)) { atomicAdd(&ctr[1], 1u); break; }       // watchdog: never hang the box
          }
          if (!(fence_mode & 16)) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
      }
      if (fence_mode & 32) copy_rows_t<true>(a + p * plane, slot, t); else copy_rows(a + p * plane, slot, t);
      if (fence_mode & 1) __threadfence();
      __syncthreads();
      if (threadIdx.x == 0) {
        if (fence_mode & 16) __hip_atomic_fetch_add(&done1[p], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else __hip_atomic_fetch_add(&done1[p], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      }
    } else {
      if (threadIdx.x == 0 && !(fence_mode & 8)) {
        unsigned spins = 0;
        while (__hip_atomic_load(&done1[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < TPP) {
          __builtin_amdgcn_s_sleep(8);
          if (++spins > (1u << 15)) { atomicAdd(&ctr[2], 1u); break; }
        }
        if (!(fence_mode & 16)) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      }
      __syncthreads();
      if (fence_mode & 1) __threadfence();
      if (fence_mode & 32) copy_cols_t<true>(slot, c + p * plane, t); else copy_cols(slot, c + p * plane, t);
      __syncthreads();
      if (threadIdx.x == 0) {
        if (fence_mode & 16) __hip_atomic_fetch_add(&done2[p], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else __hip_atomic_fetch_add(&done2[p], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
}

int main(int argc, char **argv) {
  const int planes = argc > 1 ? atoi(argv[1]) : 256;            // 256 planes x 16 MiB = 4 GiB per array
  const size_t plane = (size_t)ROWS * ROWS, n = plane * planes;
  vec4 *a, *b, *c, *ring;
  unsigned *ctr;
  CK(hipMalloc(&a, n * 16)); CK(hipMalloc(&b, n * 16)); CK(hipMalloc(&c, n * 16));
  CK(hipMalloc(&ring, plane * 16 * 64));
  CK(hipMalloc(&ctr, (16 + 2 * planes) * 4));
  CK(hipMemset(a, 0, n * 16));
  {
    std::vector<float> h(plane * 4);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)(i % 1000) * 0.001f;
    for (int p = 0; p < planes; ++p) CK(hipMemcpy(a + p * plane, h.data(), plane * 16, hipMemcpyHostToDevice));
  }
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  int cus = 0;
  CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
  setvbuf(stdout, nullptr, _IOLBF, 0);
  printf("CUs %d, array %.1f GiB, plane 16 MiB, %d tiles per plane-phase\n", cus, n * 16.0 / (1 << 30), TPP);
  auto timeit = [&](auto fn, int reps) {
    fn(); CK(hipDeviceSynchronize());
    float best = 1e9f;
    for (int r = 0; r < reps; ++r) {
      CK(hipEventRecord(e0, 0)); fn(); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    return best;
  };
  const double gb = n * 16.0 / 1e9;
  for (int grid : {cus, 2 * cus, 8192}) {
    float t = timeit([&] { hipLaunchKernelGGL(whole_rows, dim3(grid), dim3(THREADS), 0, 0, a, b, planes);
                           hipLaunchKernelGGL(whole_cols, dim3(grid), dim3(THREADS), 0, 0, b, c, planes); }, 5);
    printf("baseline two launches, grid %5d: %8.3f ms   %7.1f GB/s over 4 S\n", grid, t, 4 * gb / (t * 1e-3));
    fflush(stdout);
  }
  // reference for the result check
  vec4 *cref; CK(hipMalloc(&cref, n * 16)); CK(hipMemcpy(cref, c, n * 16, hipMemcpyDeviceToDevice));
  unsigned *dbad; CK(hipMalloc(&dbad, 4));
  std::vector<float> want(4096), got(4096);
  CK(hipMemcpy(want.data(), c + (size_t)(planes - 1) * plane + 12345, 4096 * 4, hipMemcpyDeviceToHost));
  const int dbg = argc > 2 ? atoi(argv[2]) : 0;
  for (int fence : {0 | dbg})
    for (int ringp : {8, 16}) {
      for (int lag : {4}) {
        if (lag + 2 > ringp) continue;
        for (int grid : {cus}) {
          CK(hipMemset(c, 0, n * 16));
          float best = 1e9f;
          unsigned h[8] = {0};
          for (int rep = 0; rep < 3; ++rep) {
            CK(hipMemset(ctr, 0, (16 + 2 * planes) * 4));
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(fused, dim3(grid), dim3(THREADS), 0, 0, a, ring, c, ctr, planes, ringp, lag, fence);
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
            CK(hipMemcpy(h, ctr, 32, hipMemcpyDeviceToHost));
            if (h[1] || h[2]) break;
          }
          CK(hipMemcpy(got.data(), c + (size_t)(planes - 1) * plane + 12345, 4096 * 4, hipMemcpyDeviceToHost));
          int bad = 0;
          for (int i = 0; i < 4096; ++i) bad += got[i] != want[i];
          CK(hipMemset(dbad, 0, 4));
          hipLaunchKernelGGL(diff_kernel, dim3(2048), dim3(256), 0, 0, c, cref, n, dbad);
          unsigned nb = 0; CK(hipMemcpy(&nb, dbad, 4, hipMemcpyDeviceToHost));
          bad += (int)(nb > 0x7fffffffu ? 0x7fffffff : nb);
          printf("fused fence %d ring %2d planes (%4d MiB) lag %d grid %4d: %8.3f ms   %7.1f GB/s over 4 S   %s  (tickets %u, watchdog %u/%u)\n", fence, ringp,
                 ringp * 16, lag, grid, best, 4 * gb / (best * 1e-3), bad ? "MISMATCH" : "ok (whole array)", h[0], h[1], h[2]);
          fflush(stdout);
          if (h[1] || h[2]) { printf("watchdog fired: stopping\n"); return 2; }
        }
      }
    }
  return 0;
}
