// Small 2-D planes entirely on chip: both passes of a batched 2-D complex transform over planes of n x n points (n = 32, 64) in ONE
// launch, one workgroup per plane -- rows from HBM, the plane held in LDS, columns back to HBM -- instead of two passes through
// memory.  What config C1 (64^3, the size the reference's own benchmark times: /root/reference/tests/test_speed.py:15-20) and
// fftn(axes=(1, 2)) over small images are made of: such arrays fit the caches, so a pass costs its dependent memory round trip
// (~5 us at 64^3, whatever the 4 MiB it moves), and a 3-D transform three of them; this removes one (tools/c1_probe.py).
// Both passes are the register-resident pass of fft_pow2_impl.h, called as device functions on one tile each: the row pass
// writes its tile -- the whole plane -- into LDS (rows pitched 8 entries wider: bank-conflict-free 16-lane stores), the strided pass
// reads it from there.
#include "fft_pow2_impl.h"

namespace gfft {

template <typename A, typename B>
__global__ void __launch_bounds__(A::threads)
fft_plane2d_kernel(PassDesc dA, PassDesc dB, int planes, int64_t in_plane, int64_t out_plane, double scale_b, const void *__restrict__ in,
                   void *__restrict__ out) {
  static_assert(A::threads == B::threads, "both passes of a plane run on one workgroup shape");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr size_t exch = ((A::lds > B::lds ? A::lds : B::lds) + 255) & ~(size_t)255;
  unsigned char *plane = smem + exch;
  for (int p = blockIdx.x; p < planes; p += gridDim.x) {
    A::tile(dA, static_cast<const char *>(in) + (size_t)p * in_plane, plane, smem, 0u, 1.0, NoHook());
    __syncthreads();
    B::tile(dB, plane, static_cast<char *>(out) + (size_t)p * out_plane, smem, 0u, scale_b, NoHook());
    __syncthreads();
  }
}

//                real  N   R   T  COLS   SPLIT  FLAGS        MODE      BIGTW radices       (FLAGS 8192: natural layouts; 1 / 2: non-temporal HBM side)
template <typename real, int N> struct PlaneCfg;
template <> struct PlaneCfg<double, 64> {
  typedef PassCfg<double, 64, 8, 64, false, true, 1 | 8192, MODE_C2C, false, 8, 8> Rows;
  typedef PassCfg<double, 64, 8, 64, true, true, 2 | 8 | 8192, MODE_C2C, false, 8, 8> Cols;
};
template <> struct PlaneCfg<double, 32> {
  typedef PassCfg<double, 32, 8, 32, false, true, 1 | 8192, MODE_C2C, false, 8, 4> Rows;
  typedef PassCfg<double, 32, 8, 32, true, true, 2 | 8 | 8192, MODE_C2C, false, 8, 4> Cols;
};
template <> struct PlaneCfg<float, 64> {
  typedef PassCfg<float, 64, 8, 64, false, false, 1 | 8192, MODE_C2C, false, 8, 8> Rows;
  typedef PassCfg<float, 64, 8, 64, true, false, 2 | 8 | 8192, MODE_C2C, false, 8, 8> Cols;
};
template <> struct PlaneCfg<float, 32> {
  typedef PassCfg<float, 32, 8, 32, false, false, 1 | 8192, MODE_C2C, false, 8, 4> Rows;
  typedef PassCfg<float, 32, 8, 32, true, false, 2 | 8 | 8192, MODE_C2C, false, 8, 4> Cols;
};

bool plane2d_supported(int n, int precision) { return (n == 32 || n == 64) && (precision == 4 || precision == 8); }
int plane2d_pitch(int n) { return n + 8; }      // entries between consecutive rows of the plane in LDS

template <typename real, int N>
static hipError_t launch_plane2d_one(const PassDesc &dA, const PassDesc &dB, int planes, int64_t in_plane, int64_t out_plane, const void *in,
                                     void *out, hipStream_t s) {
  typedef typename PlaneCfg<real, N>::Rows A;
  typedef typename PlaneCfg<real, N>::Cols B;
  constexpr size_t exch = ((A::lds > B::lds ? A::lds : B::lds) + 255) & ~(size_t)255;
  constexpr size_t lds = exch + (size_t)N * (N + 8) * 2 * sizeof(real);
  static_assert(lds <= 160 * 1024, "LDS budget");
  if (dA.out_os != N + 8 || dB.in_es != N + 8 || A::ntiles(dA) != 1 || B::ntiles(dB) != 1) return hipErrorInvalidValue;
  auto kern = fft_plane2d_kernel<A, B>;
  static bool attr_set[kMaxDevices] = {};
  static int cus_of[kMaxDevices] = {};
  const int dev = current_device();
  if (!attr_set[dev] && lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    attr_set[dev] = true;
  }
  if (!cus_of[dev]) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1) { (void)hipGetLastError(); n = 256; }
    cus_of[dev] = n;
  }
  // (one plane per workgroup; more planes than 4 workgroups per CU: the workgroups walk)
  const int cap = 4 * cus_of[dev];
  hipLaunchKernelGGL(kern, dim3(planes < cap ? planes : cap), dim3(A::threads), lds, s, dA, dB, planes, in_plane, out_plane, dB.scale, in, out);
  return hipGetLastError();
}

hipError_t launch_plane2d(const PassDesc &dA, const PassDesc &dB, int precision, int planes, int64_t in_plane, int64_t out_plane,
                          const void *in, void *out, hipStream_t s) {
  if (dA.n != dB.n) return hipErrorInvalidValue;
  if (precision == 8) {
    if (dA.n == 64) return launch_plane2d_one<double, 64>(dA, dB, planes, in_plane, out_plane, in, out, s);
    if (dA.n == 32) return launch_plane2d_one<double, 32>(dA, dB, planes, in_plane, out_plane, in, out, s);
  } else {
    if (dA.n == 64) return launch_plane2d_one<float, 64>(dA, dB, planes, in_plane, out_plane, in, out, s);
    if (dA.n == 32) return launch_plane2d_one<float, 32>(dA, dB, planes, in_plane, out_plane, in, out, s);
  }
  return hipErrorInvalidValue;
}

}  // namespace gfft
