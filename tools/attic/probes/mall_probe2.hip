// Developer probe (round 4): what the Infinity Cache (MALL, 256 MiB) actually holds and carries while a fused
// pass pair runs -- the questions behind the 9.7 ms of the fused [strided -> rows] launch, whose HBM minimum is 2 S:
//   1  capacity : read-only sweeps over a buffer of S MiB, repeated inside one launch: GB/s against S (where is the cliff?)
//   2  pollution: half the workgroups stream a 8 GiB copy (policy: plain / nt loads / nt stores / both) while the other
//                 half re-read a resident buffer (policy plain / sc0 sc1): does the stream evict the resident set?
//   3  hand-off : every workgroup writes a 256 KiB tile into a ring slot with sc0 sc1 stores, waits for the
//                 acknowledgements, and reads ANOTHER workgroup's previous tile back with sc0 sc1 loads: aggregate rate
// All workgroups are 1024 threads moving 16 x 16 B per thread per burst (one 256 KiB tile), as the FFT tiles do.
// Build: hipcc -O3 --offload-arch=gfx950 mall_probe2.hip -o mall_probe2
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
typedef unsigned u4 __attribute__((ext_vector_type(4)));
constexpr int THREADS = 1024, U = 16;
constexpr size_t TILE = (size_t)THREADS * U;       // 16-byte elements per tile = 256 KiB

struct SysBuf {
  __amdgpu_buffer_rsrc_t r;
  __device__ __forceinline__ explicit SysBuf(const void *base) : r(__builtin_amdgcn_make_buffer_rsrc((void *)base, 0, 0x7fffffff, 0x00020000)) {}
  template <int AUX> __device__ __forceinline__ void st(unsigned elem, u4 v) const { __builtin_amdgcn_raw_buffer_store_b128(v, r, elem * 16u, 0, AUX); }
  template <int AUX> __device__ __forceinline__ u4 ld(unsigned elem) const { return __builtin_amdgcn_raw_buffer_load_b128(r, elem * 16u, 0, AUX); }
};
// aux: 0 plain, 1 sc0, 2 nt, 16 sc1, 17 sc0 sc1 (system scope), 19 sc0 sc1 nt

template <int AUX> __device__ __forceinline__ void load_tile(const u4 *base, size_t tile, u4 *v) {
  const SysBuf sb(base + tile * TILE);
#pragma unroll
  for (int u = 0; u < U; ++u) v[u] = sb.ld<AUX>((unsigned)(u * THREADS + threadIdx.x));
}
template <int AUX> __device__ __forceinline__ void store_tile(u4 *base, size_t tile, const u4 *v) {
  const SysBuf sb(base + tile * TILE);
#pragma unroll
  for (int u = 0; u < U; ++u) sb.st<AUX>((unsigned)(u * THREADS + threadIdx.x), v[u]);
}

// 1: read a buffer of `tiles` tiles `passes` times; workgroup b reads tiles b, b + grid, ... (each XCD sees 1/8 of them)
template <int AUX>
__global__ void __launch_bounds__(THREADS) read_resident(const u4 *buf, size_t tiles, int passes, u4 *sink) {
  u4 acc = {0, 0, 0, 0};
  for (int p = 0; p < passes; ++p)
    for (size_t t = blockIdx.x; t < tiles; t += gridDim.x) {
      u4 v[U];
      load_tile<AUX>(buf, t, v);
#pragma unroll
      for (int u = 0; u < U; ++u) acc ^= v[u];
    }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x1234567u) *sink = acc;
}

// 2: even workgroups copy a big array tile by tile (LD / ST policies), odd workgroups re-read the resident buffer with
// policy RD until the copy is done; every workgroup counts what it moved
template <int LD, int ST, int RD>
__global__ void __launch_bounds__(THREADS)
pollute(const u4 *src, u4 *dst, size_t stream_tiles, const u4 *res, size_t res_tiles, unsigned *done, unsigned long long *moved, u4 *sink) {
  const unsigned half = gridDim.x / 2, me = blockIdx.x / 2;
  u4 acc = {0, 0, 0, 0};
  if ((blockIdx.x & 1) == 0) {
    for (size_t t = me; t < stream_tiles; t += half) {
      u4 v[U];
      load_tile<LD>(src, t, v);
      store_tile<ST>(dst, t, v);
    }
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(done, 1u);
  } else {
    unsigned long long n = 0;
    size_t t = me;
    for (;;) {
      u4 v[U];
      load_tile<RD>(res, t, v);
#pragma unroll
      for (int u = 0; u < U; ++u) acc ^= v[u];
      ++n;
      t += half;
      if (t >= res_tiles) t -= res_tiles / half * half;
      if (t >= res_tiles) t = me;
      if ((n & 7) == 0) {
        __shared__ unsigned stop;
        if (threadIdx.x == 0) stop = __hip_atomic_load(done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= half;
        __syncthreads();
        const bool s = stop != 0;
        __syncthreads();
        if (s) break;
      }
    }
    if (threadIdx.x == 0) atomicAdd(moved, n);
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x1234567u) *sink = acc;
}

// 3: ring hand-off at full tilt, no HBM stream: workgroup b writes tile (round, b) of a ring of `slots` x grid tiles with
// write-through stores, waits for the acknowledgements, raises its flag, then reads the tile workgroup (b + shift) % grid
// wrote one round EARLIER (certainly complete: a grid-wide counter orders the rounds loosely) back at system scope.
template <int ST, int LD>
__global__ void __launch_bounds__(THREADS) handoff(u4 *ring, int slots, int rounds, int shift, u4 *sink) {
  u4 acc = {0, 0, 0, 0};
  u4 v[U];
#pragma unroll
  for (int u = 0; u < U; ++u) v[u] = u4{(unsigned)blockIdx.x, (unsigned)u, threadIdx.x, 1u};
  for (int r = 0; r < rounds; ++r) {
    store_tile<ST>(ring, (size_t)(r % slots) * gridDim.x + blockIdx.x, v);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (r > 0) {
      u4 w[U];
      load_tile<LD>(ring, (size_t)((r - 1) % slots) * gridDim.x + (blockIdx.x + shift) % gridDim.x, w);
#pragma unroll
      for (int u = 0; u < U; ++u) acc ^= w[u];
    }
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x1234567u) *sink = acc;
}

static hipEvent_t e0, e1;
template <typename F> float timeit(F fn, int reps) {
  fn();
  CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int r = 0; r < reps; ++r) {
    CK(hipEventRecord(e0, 0));
    fn();
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    best = std::min(best, ms);
  }
  CK(hipGetLastError());
  return best;
}

int main() {
  setvbuf(stdout, nullptr, _IOLBF, 0);
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  int cus = 0;
  CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
  const size_t big = (size_t)8 << 30, resmax = (size_t)1 << 30;
  u4 *src, *dst, *res, *sink;
  unsigned *done;
  unsigned long long *moved;
  CK(hipMalloc(&src, big)); CK(hipMalloc(&dst, big)); CK(hipMalloc(&res, resmax)); CK(hipMalloc(&sink, 64));
  CK(hipMalloc(&done, 64)); CK(hipMalloc(&moved, 64));
  CK(hipMemset(src, 1, big)); CK(hipMemset(dst, 2, big)); CK(hipMemset(res, 3, resmax));
  printf("%d CUs; tiles of 256 KiB, 1024-thread workgroups\n", cus);

  printf("\n== 1 capacity: read-only sweeps of a resident buffer, 24 passes in one launch, grid = CUs\n");
  for (int mib : {32, 64, 96, 128, 160, 192, 224, 240, 256, 288, 320, 384, 512, 1024}) {
    const size_t tiles = ((size_t)mib << 20) / (TILE * 16);
    const int passes = 24;
    const float t0 = timeit([&] { hipLaunchKernelGGL(read_resident<0>, dim3(cus), dim3(THREADS), 0, 0, res, tiles, passes, sink); }, 3);
    const float t1 = timeit([&] { hipLaunchKernelGGL(read_resident<17>, dim3(cus), dim3(THREADS), 0, 0, res, tiles, passes, sink); }, 3);
    const float t2 = timeit([&] { hipLaunchKernelGGL(read_resident<2>, dim3(cus), dim3(THREADS), 0, 0, res, tiles, passes, sink); }, 3);
    const double b = (double)passes * mib * 1048576.0;
    printf("  %5d MiB: plain %7.0f GB/s   sc0 sc1 %7.0f GB/s   nt %7.0f GB/s\n", mib, b / t0 / 1e6, b / t1 / 1e6, b / t2 / 1e6);
  }

  printf("\n== 2 pollution: 128 workgroups copy 8 GiB (HBM stream) while 128 re-read a resident buffer\n");
  auto run2 = [&](const char *name, auto kern, int res_mib) {
    const size_t st = big / (TILE * 16), rt = ((size_t)res_mib << 20) / (TILE * 16);
    unsigned long long h = 0;
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
      CK(hipMemset(done, 0, 64)); CK(hipMemset(moved, 0, 64));
      CK(hipEventRecord(e0, 0));
      hipLaunchKernelGGL(kern, dim3(cus), dim3(THREADS), 0, 0, src, dst, st, res, rt, done, moved, sink);
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best) { best = ms; CK(hipMemcpy(&h, moved, 8, hipMemcpyDeviceToHost)); }
    }
    printf("  %-44s resident %3d MiB: stream %6.0f GB/s, resident reads %6.0f GB/s (%.2f ms)\n", name, res_mib, 2.0 * big / best / 1e6,
           (double)h * TILE * 16 / best / 1e6, best);
  };
  for (int res_mib : {64, 128, 192}) {
    run2("stream plain / plain, resident plain", pollute<0, 0, 0>, res_mib);
    run2("stream nt loads / plain stores, resident plain", pollute<2, 0, 0>, res_mib);
    run2("stream plain loads / nt stores, resident plain", pollute<0, 2, 0>, res_mib);
    run2("stream nt / nt, resident plain", pollute<2, 2, 0>, res_mib);
    run2("stream nt / nt, resident sc0 sc1", pollute<2, 2, 17>, res_mib);
    run2("stream plain / plain, resident sc0 sc1", pollute<0, 0, 17>, res_mib);
    run2("stream sc1 nt / sc1 nt, resident sc0 sc1", pollute<18, 18, 17>, res_mib);
  }

  printf("\n== 3 hand-off at full tilt (no HBM stream): write-through a 256 KiB tile, wait for the acks, read a neighbour's back\n");
  for (int slots : {2, 3}) {
    const int rounds = 200;
    for (int shift : {1, 8, 37}) {
      const float t17 = timeit([&] { hipLaunchKernelGGL((handoff<17, 17>), dim3(cus), dim3(THREADS), 0, 0, src, slots, rounds, shift, sink); }, 3);
      const float t0 = timeit([&] { hipLaunchKernelGGL((handoff<0, 0>), dim3(cus), dim3(THREADS), 0, 0, src, slots, rounds, shift, sink); }, 3);
      const float t16 = timeit([&] { hipLaunchKernelGGL((handoff<16, 16>), dim3(cus), dim3(THREADS), 0, 0, src, slots, rounds, shift, sink); }, 3);
      const double b = 2.0 * rounds * cus * TILE * 16;
      printf("  ring of %d x %d MiB, reader shift %2d: sc0 sc1 %6.0f GB/s (%.1f us per round)   plain %6.0f GB/s   sc1 %6.0f GB/s\n", slots, (int)(cus * TILE * 16 >> 20), shift,
             b / t17 / 1e6, t17 * 1e3 / rounds, b / t0 / 1e6, b / t16 / 1e6);
    }
  }
  return 0;
}
