// Developer probe (round 4): does the streaming rate belong to the BUFFER?  profiles/r04_copy_sweep.txt shows the same copy
// between freshly allocated 16 GiB pairs at 5.4 - 6.5 TB/s.  Allocate K buffers of 16 GiB, measure each one's read-only
// and write-only rate and every ordered pair's copy rate with the best copy geometry: if a buffer is slow or fast by
// itself, a planner can choose among candidates at plan time (as FFTW_MEASURE chooses among algorithms).
// Build: hipcc -O3 --offload-arch=gfx950 placement_probe.hip -o placement_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
typedef unsigned u4 __attribute__((ext_vector_type(4)));
constexpr int THREADS = 1024, U = 8;
template <int MODE>   // 0 copy, 1 read, 2 write
__global__ void __launch_bounds__(THREADS) k(const u4 *__restrict__ src, u4 *__restrict__ dst, long long n, u4 *sink) {
  const long long chunk = (long long)U * THREADS;
  u4 acc = {0, 0, 0, 0};
  for (long long base = (long long)blockIdx.x * chunk; base + chunk <= n; base += (long long)gridDim.x * chunk) {
    u4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = MODE == 2 ? u4{(unsigned)base, 1u, 2u, 3u} : __builtin_nontemporal_load(src + base + u * THREADS + threadIdx.x);
#pragma unroll
    for (int u = 0; u < U; ++u) { if (MODE == 1) acc ^= v[u]; else __builtin_nontemporal_store(v[u], dst + base + u * THREADS + threadIdx.x); }
  }
  if (MODE == 1 && (acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x1234567u) *sink = acc;
}
static hipEvent_t e0, e1;
template <int MODE> float run(const u4 *a, u4 *b, long long n, u4 *sink) {
  float best = 1e30f;
  for (int r = 0; r < 4; ++r) {
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k<MODE>, dim3(4096), dim3(THREADS), 0, 0, a, b, n, sink);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (r) best = std::min(best, ms);
  }
  return best;
}
int main(int argc, char **argv) {
  const int K = argc > 1 ? atoi(argv[1]) : 8;
  const size_t bytes = (size_t)16 << 30;
  const long long n = bytes / 16;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<u4 *> buf(K);
  u4 *sink; CK(hipMalloc(&sink, 64));
  for (int i = 0; i < K; ++i) { CK(hipMalloc(&buf[i], bytes)); CK(hipMemset(buf[i], i + 1, bytes)); }
  printf("%d buffers of 16 GiB\n", K);
  for (int i = 0; i < K; ++i)
    printf("buffer %d at %p: read %7.0f GB/s  write %7.0f GB/s\n", i, (void *)buf[i], bytes / run<1>(buf[i], nullptr, n, sink) / 1e6, bytes / run<2>(nullptr, buf[i], n, sink) / 1e6);
  printf("copy src (row) -> dst (column), GB/s over 2 x 16 GiB:\n      ");
  for (int j = 0; j < K; ++j) printf("%7d", j);
  printf("\n");
  for (int i = 0; i < K; ++i) {
    printf("  %2d: ", i);
    for (int j = 0; j < K; ++j) {
      if (i == j) { printf("      -"); continue; }
      printf("%7.0f", 2.0 * bytes / run<0>(buf[i], buf[j], n, sink) / 1e6);
    }
    printf("\n");
  }
  // again, to see whether the figures are stable
  printf("second pass, copy i -> i+1: ");
  for (int i = 0; i + 1 < K; ++i) printf("%7.0f", 2.0 * bytes / run<0>(buf[i], buf[i + 1], n, sink) / 1e6);
  printf("\n");
  return 0;
}
