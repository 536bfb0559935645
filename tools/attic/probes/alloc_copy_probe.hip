#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void copyk(const double2* __restrict__ s, double2* __restrict__ d, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
  for (; i + 3 * st < n; i += 4 * st) {
    double2 a = s[i], b = s[i + st], c = s[i + 2 * st], e = s[i + 3 * st];
    d[i] = a; d[i + st] = b; d[i + 2 * st] = c; d[i + 3 * st] = e;
  }
  for (; i < n; i += st) d[i] = s[i];
}
int main() {
  size_t bytes = (size_t)16 << 30, n = bytes / 16;
  for (int mode = 0; mode < 3; ++mode) {
    void *a = nullptr, *b = nullptr;
    hipError_t e1, e2;
    if (mode == 0) { e1 = hipMalloc(&a, bytes); e2 = hipMalloc(&b, bytes); }
    else if (mode == 1) { e1 = hipExtMallocWithFlags(&a, bytes, hipDeviceMallocContiguous); e2 = hipExtMallocWithFlags(&b, bytes, hipDeviceMallocContiguous); }
    else { e1 = hipExtMallocWithFlags(&a, bytes, hipDeviceMallocUncached); e2 = hipExtMallocWithFlags(&b, bytes, hipDeviceMallocUncached); }
    if (e1 != hipSuccess || e2 != hipSuccess) { printf("mode %d alloc failed: %s %s\n", mode, hipGetErrorString(e1), hipGetErrorString(e2)); (void)hipGetLastError(); continue; }
    hipMemset(a, 1, bytes);
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    for (int g : {2048, 8192, 65536}) {
      copyk<<<g, 256>>>((double2*)a, (double2*)b, n);
      hipEventRecord(s);
      for (int it = 0; it < 5; ++it) copyk<<<g, 256>>>((double2*)a, (double2*)b, n);
      hipEventRecord(e); hipEventSynchronize(e);
      float ms; hipEventElapsedTime(&ms, s, e);
      printf("mode %d (%s) grid %6d: %.3f ms per 2x16 GiB  %.1f GB/s\n", mode, mode == 0 ? "hipMalloc" : mode == 1 ? "contiguous" : "uncached", g, ms / 5, 2.0 * bytes * 5 / ms / 1e6);
    }
    hipFree(a); hipFree(b);
  }
  return 0;
}
