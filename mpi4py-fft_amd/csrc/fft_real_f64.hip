// fp64 packed-real row kernels: r2c / c2r of even length 2N along a contiguous axis as ONE complex
// transform of length N on the line read as complex pairs, plus the Hermitian pass in registers
// (fft_pow2_impl.h, MODE_R2C_H / MODE_C2R_H).  Replaces the full-length transform with a
// zero-imaginary load adapter for these lines: half the butterflies, 16-byte loads.
//
// Plan table: N = complex length, R = entries per thread, T = lines per workgroup.
#include "fft_pow2_impl.h"

namespace gfft {

// (plain, truncating-store and zero-padding-load instantiations of one plan: half_launch picks by d.tr_dir)
#define H64(MODE, N, R, T, ...) half_launch<double, MODE, N, R, T, true, __VA_ARGS__>(d, in, out, s)

template <int MODE>
static hipError_t launch_half_f64(const PassDesc &d, int variant, const void *in, void *out, hipStream_t s) {
  switch (d.n) {
    case 16: return H64(MODE, 16, 4, 16, 4, 4);
    case 32: return H64(MODE, 32, 8, 16, 8, 4);
    case 64: return H64(MODE, 64, 8, 8, 8, 8);
    case 128: return H64(MODE, 128, 8, 4, 8, 8, 2);
    case 256: return H64(MODE, 256, 8, 8, 8, 8, 4);
    // (512 ... 2048: R = 16 measured 5-10 % faster than R = 8 in both directions despite 157 VGPRs;
    // variants 2 / 3 keep the R = 8 plans for A/B runs)
    case 512:
      switch (variant) {
        default: return H64(MODE, 512, 16, 8, 16, 8, 4);
#ifdef GFFT_VARIANTS
        case 2: return H64(MODE, 512, 8, 4, 8, 8, 8);
        case 3: return H64(MODE, 512, 8, 8, 8, 8, 8);
#endif
      }
    case 1024:
      switch (variant) {
        default: return H64(MODE, 1024, 16, 4, 16, 16, 4);
#ifdef GFFT_VARIANTS
        case 2: return H64(MODE, 1024, 8, 2, 8, 8, 8, 2);
        case 3: return H64(MODE, 1024, 8, 4, 8, 8, 8, 2);
#endif
      }
    case 2048:
      switch (variant) {
        default: return H64(MODE, 2048, 16, 2, 16, 16, 8);
#ifdef GFFT_VARIANTS
        case 2: return H64(MODE, 2048, 8, 1, 8, 8, 8, 4);
        case 3: return H64(MODE, 2048, 8, 2, 8, 8, 8, 4);
#endif
      }
    case 4096: return H64(MODE, 4096, 16, 1, 16, 16, 16);
  }
  return hipErrorInvalidValue;
}

bool real_half_supported(int n_complex) { return n_complex >= 16 && n_complex <= 4096 && (n_complex & (n_complex - 1)) == 0; }

hipError_t launch_real_half_f64(const PassDesc &d, int variant, const void *in, void *out, hipStream_t s) {
  if (d.mode == MODE_R2C_H) return launch_half_f64<MODE_R2C_H>(d, variant, in, out, s);
  if (d.mode == MODE_C2R_H) return launch_half_f64<MODE_C2R_H>(d, variant, in, out, s);
  return hipErrorInvalidValue;
}

}  // namespace gfft
