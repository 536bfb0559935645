// fp32 packed-real row kernels (see fft_real_f64.hip): r2c / c2r of even length 2N along a
// contiguous axis as one complex64 transform of length N plus the Hermitian pass in registers.
// fp32 exchanges whole complex values through LDS, so the mirror entries are combined as they are
// read (no extra registers) and R = 16 plans stay under 128 VGPRs.
#include "fft_pow2_impl.h"

namespace gfft {

// (R = 16 plans are held to 128 VGPRs = 4 waves per SIMD: unconstrained, the backward kernels
// keep all mirror entries and twiddles in flight at once and take ~170)
// (plain, truncating-store and zero-padding-load instantiations of one plan: half_launch picks by d.tr_dir)
#define H32(MODE, N, R, T, ...) half_launch<float, MODE, N, R, T, false, __VA_ARGS__>(d, in, out, s)

template <int MODE>
static hipError_t launch_half_f32(const PassDesc &d, int variant, const void *in, void *out, hipStream_t s) {
  switch (d.n) {
    case 16: return H32(MODE, 16, 4, 16, 4, 4);
    case 32: return H32(MODE, 32, 8, 16, 8, 4);
    case 64: return H32(MODE, 64, 8, 8, 8, 8);
    case 128: return H32(MODE, 128, 8, 4, 8, 8, 2);
    case 256: return H32(MODE, 256, 16, 4, 16, 16);
    // measured on (1024,1024,1024) / (512,1024,2048) / (256,1024,4096) f32 (tools/real_probe.py):
    // forward likes R = 16 everywhere; backward R = 16 takes ~170 VGPRs and only wins from N = 1024
    case 512:
      switch (variant) {
        default:
          if (MODE == MODE_C2R_H) return H32(MODE, 512, 8, 4, 8, 8, 8);
          return H32(MODE, 512, 16, 8, 16, 8, 4);
#ifdef GFFT_VARIANTS
        case 2: return H32(MODE, 512, 8, 8, 8, 8, 8);
        case 3: return H32(MODE, 512, 8, 1, 8, 8, 8);
#endif
      }
    case 1024:
      switch (variant) {
        // Round 6: 4 rows per workgroup (256 threads) instead of 1 (a wave): config C5's first stage (512,1024,2048) r2c 1.764 / 1.797 -> 1.663 /
        // 1.726 ms on two boxes, c2r 1.715 / 1.711 -> 1.688 / 1.675 ms; 2 rows level with 4, 8 rows level with 1.  ONE exchange (32 values per
        // thread, radices 32 x 32, 2 or 8 rows) -- what pays on the strided tiles -- LOSES 2-4 % here: a row's exchanges stay inside its wave and
        // cost no barrier, the 190-220 VGPRs cost occupancy.  Other lengths (1024 and 4096 reals per row): every row count within 2 % of the
        // table's (tools/real_rows_variant_probe.py, profiles/r06_real_rows_probe.txt).  Non-temporal loads and stores -- a gain of 4-10 % on the
        // COMPLEX row passes -- are level to worse here (1024 reals per row: r2c 1.744 -> 1.960 ms).  variant 1 = one row per wave.
        default: return H32(MODE, 1024, 16, 4, 16, 16, 4);
        case 1: return H32(MODE, 1024, 16, 1, 16, 16, 4);
#ifdef GFFT_VARIANTS
        case 3: return H32(MODE, 1024, 8, 4, 8, 8, 8, 2);
#endif
      }
    case 2048:
      switch (variant) {
        default: return H32(MODE, 2048, 16, 1, 16, 16, 8);
#ifdef GFFT_VARIANTS
        case 2: return H32(MODE, 2048, 16, 2, 16, 16, 8);
        case 3: return H32(MODE, 2048, 8, 2, 8, 8, 8, 4);
#endif
      }
    case 4096: return H32(MODE, 4096, 16, 1, 16, 16, 16);
  }
  return hipErrorInvalidValue;
}

hipError_t launch_real_half_f32(const PassDesc &d, int variant, const void *in, void *out, hipStream_t s) {
  if (d.mode == MODE_R2C_H) return launch_half_f32<MODE_R2C_H>(d, variant, in, out, s);
  if (d.mode == MODE_C2R_H) return launch_half_f32<MODE_C2R_H>(d, variant, in, out, s);
  return hipErrorInvalidValue;
}

}  // namespace gfft
