// Developer probe (round 4): what is the streaming ceiling of THIS box?  bench.py's copy probe
// (csrc/pack.hip copy_kernel: 256 threads, 4 x 16 B per thread in flight, 8192 workgroups) reaches 5.3-5.6 TB/s at
// 16 GiB; /opt/skills/guides/MI355X_MICROARCH.md:35 quotes 6.29 TB/s for a float4 copy.  Sweep the copy over
//   mode      copy (read + write), read only, write only, in place (read and write the same lines)
//   U         16-byte accesses in flight per thread: 1 2 4 8 16
//   policy    plain / non-temporal loads x plain / non-temporal stores
//   geometry  workgroups = 256 CUs x {1 2 4 8 16 32 64}, 256 / 512 / 1024 threads
//   order     grid-stride chunks (neighbouring workgroups on neighbouring chunks) or one contiguous share per workgroup
//   size      1 ... 16 GiB per buffer
// and print GB/s (bytes moved / time: 2 x size for a copy).  Build: hipcc -O3 --offload-arch=gfx950 copy_sweep.hip -o copy_sweep
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef unsigned u4 __attribute__((ext_vector_type(4)));
enum { COPY = 0, READ = 1, WRITE = 2 };

template <bool NT> __device__ __forceinline__ u4 ld(const u4 *p) { return NT ? __builtin_nontemporal_load(p) : *p; }
template <bool NT> __device__ __forceinline__ void st(u4 *p, u4 v) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; }

// n = number of 16-byte elements (a multiple of U * blockDim.x * gridDim.x in every call below)
template <int MODE, int U, bool NTL, bool NTS, bool BLOCKED>
__global__ void sweep_kernel(const u4 *__restrict__ src, u4 *__restrict__ dst, long long n, u4 *sink, int passes) {
  const long long chunk = (long long)U * blockDim.x;
  const long long chunks = n / chunk, per_block = chunks / gridDim.x;
  u4 acc = {0, 0, 0, 0};
  for (int pass = 0; pass < passes; ++pass)
  for (long long k = 0; k < per_block; ++k) {
    // (repeated passes: every pass deals the chunks to other workgroups -- and other XCDs --, so that a re-read is
    // served by neither the CU's L1 nor its XCD's L2 but by the memory side)
    const long long bb = passes > 1 ? (blockIdx.x + (long long)pass * 7919) % gridDim.x : blockIdx.x;
    const long long c = BLOCKED ? bb * per_block + k : k * gridDim.x + bb;
    const long long base = c * chunk + threadIdx.x;
    u4 v[U];
    if (MODE != WRITE) {
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = ld<NTL>(src + base + (long long)u * blockDim.x);
    } else {
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = u4{(unsigned)base, (unsigned)u, 0u, 1u};
    }
    if (MODE != READ) {
#pragma unroll
      for (int u = 0; u < U; ++u) st<NTS>(dst + base + (long long)u * blockDim.x, v[u]);
    } else {
#pragma unroll
      for (int u = 0; u < U; ++u) acc ^= v[u];
    }
  }
  if (MODE == READ && (acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) *sink = acc;     // (keeps the loads alive)
}

struct Cfg { int mode, U, ntl, nts, blocked, wgs, threads; };
typedef void (*kern_t)(const u4 *, u4 *, long long, u4 *, int);

template <int MODE, int U, bool BLOCKED> kern_t pick_pol(int ntl, int nts) {
  if (ntl) return nts ? sweep_kernel<MODE, U, true, true, BLOCKED> : sweep_kernel<MODE, U, true, false, BLOCKED>;
  return nts ? sweep_kernel<MODE, U, false, true, BLOCKED> : sweep_kernel<MODE, U, false, false, BLOCKED>;
}
template <int MODE, bool BLOCKED> kern_t pick_u(int U, int ntl, int nts) {
  switch (U) {
    case 1: return pick_pol<MODE, 1, BLOCKED>(ntl, nts);
    case 2: return pick_pol<MODE, 2, BLOCKED>(ntl, nts);
    case 4: return pick_pol<MODE, 4, BLOCKED>(ntl, nts);
    case 8: return pick_pol<MODE, 8, BLOCKED>(ntl, nts);
    default: return pick_pol<MODE, 16, BLOCKED>(ntl, nts);
  }
}
kern_t pick(const Cfg &c) {
  if (c.blocked) return c.mode == COPY ? pick_u<COPY, true>(c.U, c.ntl, c.nts) : (c.mode == READ ? pick_u<READ, true>(c.U, c.ntl, c.nts) : pick_u<WRITE, true>(c.U, c.ntl, c.nts));
  return c.mode == COPY ? pick_u<COPY, false>(c.U, c.ntl, c.nts) : (c.mode == READ ? pick_u<READ, false>(c.U, c.ntl, c.nts) : pick_u<WRITE, false>(c.U, c.ntl, c.nts));
}

static hipEvent_t e0, e1;
static u4 *sink;

// best of `reps` timed launches (ms), after one warm-up
static int g_passes = 1;
float run(const Cfg &c, const u4 *src, u4 *dst, long long n, int reps) {
  kern_t k = pick(c);
  const long long quantum = (long long)c.U * c.threads * c.wgs;
  const long long nn = n / quantum * quantum;
  float best = 1e30f;
  for (int r = 0; r <= reps; ++r) {
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k, dim3(c.wgs), dim3(c.threads), 0, 0, src, dst, nn, sink, g_passes);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (r > 0) best = std::min(best, ms);
  }
  CK(hipGetLastError());
  return best / g_passes;
}

const char *mode_name(int m) { return m == COPY ? "copy" : (m == READ ? "read" : "write"); }

int main(int argc, char **argv) {
  const long long gib = argc > 1 ? atoll(argv[1]) : 16;
  const size_t bytes = (size_t)gib << 30;
  const long long n = (long long)(bytes / 16);
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  size_t fr = 0, tot = 0;
  CK(hipMemGetInfo(&fr, &tot));
  printf("device %s, %d CUs, %.1f GiB free; buffers of %lld GiB\n", prop.name, prop.multiProcessorCount, fr / 1073741824.0, gib);
  u4 *a, *b;
  CK(hipMalloc(&a, bytes));
  CK(hipMalloc(&b, bytes));
  CK(hipMalloc(&sink, 64));
  CK(hipMemset(a, 1, bytes));
  CK(hipMemset(b, 2, bytes));
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int cus = prop.multiProcessorCount;

  // reference points: the runtime's own device-to-device copy, and bench.py's probe geometry
  {
    float best = 1e30f;
    for (int r = 0; r < 4; ++r) {
      CK(hipEventRecord(e0, 0));
      CK(hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0));
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (r) best = std::min(best, ms);
    }
    printf("hipMemcpyAsync D2D                         %8.3f ms  %7.0f GB/s\n", best, 2.0 * bytes / best / 1e6);
    Cfg ref{COPY, 4, 0, 0, 0, 8192, 256};
    float ms = run(ref, a, b, n, 3);
    printf("bench.py probe geometry (U4 256thr 8192wg) %8.3f ms  %7.0f GB/s\n", ms, 2.0 * bytes / ms / 1e6);
  }

  struct Res { Cfg c; float ms; double gbs; };
  std::vector<Res> all;
  // buffers that stay inside the 256 MiB Infinity Cache: what the fabric between the L2s and the memory side carries
  // when HBM is not in the way (each configuration runs 16 passes over the buffers inside one launch)
  if (argc > 2) {
    g_passes = 16;
    for (long long mib : {64LL, 96LL, 128LL}) {
      const long long nn = (mib << 20) / 16;
      for (int mode = 0; mode < 3; ++mode) {
        std::vector<Res> res;
        for (int U : {2, 4, 8})
          for (int pol = 0; pol < 4; ++pol) {
            const int ntl = pol & 1, nts = pol >> 1;
            if (mode == READ && nts) continue;
            if (mode == WRITE && ntl) continue;
            for (int m : {1, 2, 4, 8})
              for (int t : {256, 512, 1024}) {
                if ((long long)U * t * cus * m > nn || nn % ((long long)U * t * cus * m)) continue;
                Cfg c{mode, U, ntl, nts, 0, cus * m, t};
                const float ms = run(c, a, b, nn, 2);
                res.push_back({c, ms, (mode == COPY ? 2.0 : 1.0) * (double)(mib << 20) / ms / 1e6});
              }
          }
        std::sort(res.begin(), res.end(), [](const Res &x, const Res &y) { return x.gbs > y.gbs; });
        printf("cache-resident %s, %lld MiB per buffer: best", mode_name(mode), mib);
        for (size_t i = 0; i < 3 && i < res.size(); ++i)
          printf("  [U%d ntl%d nts%d %dwg %dthr: %.0f GB/s]", res[i].c.U, res[i].c.ntl, res[i].c.nts, res[i].c.wgs, res[i].c.threads, res[i].gbs);
        printf("  worst %.0f GB/s\n", res.back().gbs);
      }
    }
    g_passes = 1;
    if (argc > 3) return 0;
  }
  const int Us[] = {1, 2, 4, 8, 16};
  const int mults[] = {1, 2, 4, 8, 16, 32, 64};
  const int thr[] = {256, 512, 1024};
  for (int mode = 0; mode < 3; ++mode) {
    std::vector<Res> res;
    for (int U : Us)
      for (int pol = 0; pol < 4; ++pol) {
        const int ntl = pol & 1, nts = pol >> 1;
        if (mode == READ && nts) continue;
        if (mode == WRITE && ntl) continue;
        for (int blocked = 0; blocked < 2; ++blocked)
          for (int m : mults)
            for (int t : thr) {
              if ((long long)m * t > 32 * 1024) continue;      // (more than 2048 threads per CU wanted: only grid-stride tails differ)
              Cfg c{mode, U, ntl, nts, blocked, cus * m, t};
              const float ms = run(c, a, b, n, 2);
              const double moved = (mode == COPY ? 2.0 : 1.0) * bytes;
              res.push_back({c, ms, moved / ms / 1e6});
            }
      }
    std::sort(res.begin(), res.end(), [](const Res &x, const Res &y) { return x.gbs > y.gbs; });
    printf("\n== %s, %lld GiB: %zu configurations; best 12, then worst 3\n", mode_name(mode), gib, res.size());
    printf("   U ntl nts blocked   wgs thr      ms     GB/s\n");
    for (size_t i = 0; i < res.size(); ++i)
      if (i < 12 || i + 3 >= res.size())
        printf("  %2d  %d   %d     %d    %5d %4d %8.3f %8.0f\n", res[i].c.U, res[i].c.ntl, res[i].c.nts, res[i].c.blocked, res[i].c.wgs, res[i].c.threads, res[i].ms, res[i].gbs);
    // marginals: best GB/s per value of each knob
    auto marg = [&](const char *name, auto key, std::vector<int> vals) {
      printf("  best by %-8s", name);
      for (int v : vals) {
        double b2 = 0;
        for (auto &r : res) if (key(r.c) == v) b2 = std::max(b2, r.gbs);
        printf("  %d: %.0f", v, b2);
      }
      printf("\n");
    };
    marg("U", [](const Cfg &c) { return c.U; }, {1, 2, 4, 8, 16});
    marg("ntl", [](const Cfg &c) { return c.ntl; }, {0, 1});
    marg("nts", [](const Cfg &c) { return c.nts; }, {0, 1});
    marg("blocked", [](const Cfg &c) { return c.blocked; }, {0, 1});
    marg("wgs/CU", [&](const Cfg &c) { return c.wgs / cus; }, {1, 2, 4, 8, 16, 32, 64});
    marg("threads", [](const Cfg &c) { return c.threads; }, {256, 512, 1024});
    if (!res.empty()) all.push_back(res[0]);
  }

  // the best copy configuration over sizes, out of place and in place (dst == src: what an in-place pass does)
  if (!all.empty()) {
    const Cfg best = all[0].c;
    printf("\n== best copy configuration (U %d ntl %d nts %d blocked %d wgs %d thr %d) over sizes; bench.py geometry beside it\n", best.U, best.ntl, best.nts,
           best.blocked, best.wgs, best.threads);
    for (long long mib : {64LL, 128LL, 256LL, 512LL, 1024LL, 2048LL, 4096LL, 8192LL, 16384LL}) {
      if ((size_t)mib << 20 > bytes) break;
      const long long nn = (mib << 20) / 16;
      Cfg ref{COPY, 4, 0, 0, 0, (int)std::min<long long>(8192, nn / 1024), 256};
      Cfg bst = best;
      while ((long long)bst.U * bst.threads * bst.wgs > nn && bst.wgs > cus) bst.wgs /= 2;
      const float ms = run(bst, a, b, nn, 4), ms_in = run(bst, a, a, nn, 4), ms_ref = run(ref, a, b, nn, 4);
      printf("  %6lld MiB: out of place %8.4f ms %7.0f GB/s   in place %8.4f ms %7.0f GB/s   bench geometry %8.4f ms %7.0f GB/s\n", mib, ms,
             2.0 * (mib << 20) / ms / 1e6, ms_in, 2.0 * (mib << 20) / ms_in / 1e6, ms_ref, 2.0 * (mib << 20) / ms_ref / 1e6);
    }
    // placement: the same copy between buffers allocated afresh (physical placement moves the headline by several per cent)
    printf("\n== best copy configuration, 16 GiB-class buffers re-allocated 4 times\n");
    for (int r = 0; r < 4; ++r) {
      u4 *c2, *d2, *hole;
      CK(hipMalloc(&hole, (size_t)(r + 1) << 28));
      CK(hipMalloc(&c2, bytes));
      CK(hipMalloc(&d2, bytes));
      CK(hipMemset(c2, 3, bytes));
      const float ms = run(best, c2, d2, n, 3);
      printf("  allocation %d: %8.3f ms %7.0f GB/s\n", r, ms, 2.0 * bytes / ms / 1e6);
      CK(hipFree(c2)); CK(hipFree(d2)); CK(hipFree(hole));
    }
  }
  return 0;
}
