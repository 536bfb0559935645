// fp32 (complex64) one-pass kernels for lengths 3 x 5 x 2^k, 3^2 x 5 x 2^k, 3 x 5^2 x 2^k (240 ... 3840): see
// fft_mixv_f64.hip.  A complex64 is 8 bytes, so the strided kernels take up to 32 values per thread (the radix-15 stage
// keeps 30 of them) and 32 adjacent columns where the length divides by both 30 and 32.
#include "fft_pow2_impl.h"

namespace gfft {

#define X32(N, R, T, COLS, MINW, ...) \
  launch_pow2_inst<float, N, R, T, COLS, true, MINW, 8, __VA_ARGS__>(d, in, out, s)

hipError_t launch_mixv_f32(const PassDesc &d, bool cols, const void *in, void *out, hipStream_t s) {
  if (d.mode != MODE_C2C) return hipErrorInvalidValue;
  if (!cols) {
    switch (d.n) {
      case 240: return X32(240, 16, 16, false, 1, 15, 16);
      case 480: return X32(480, 16, 8, false, 1, 15, 16, 2);
      case 960: return X32(960, 16, 4, false, 1, 15, 16, 4);
      case 1920: return X32(1920, 16, 2, false, 1, 15, 16, 8);
      case 3840: return X32(3840, 16, 1, false, 1, 15, 16, 16);
      case 720: return X32(720, 16, 8, false, 1, 15, 3, 16);
      case 1440: return X32(1440, 16, 4, false, 1, 15, 3, 16, 2);
      case 2880: return X32(2880, 16, 2, false, 1, 15, 3, 16, 4);
      case 1200: return X32(1200, 16, 4, false, 1, 15, 5, 16);
      case 2400: return X32(2400, 16, 2, false, 1, 15, 5, 16, 2);
      // 7 x 2^k: the radix-7 stage keeps 14 of the 16 values
      case 112: return X32(112, 16, 16, false, 1, 7, 16);
      case 224: return X32(224, 16, 16, false, 1, 7, 16, 2);
      case 448: return X32(448, 16, 8, false, 1, 7, 16, 4);
      case 896: return X32(896, 16, 4, false, 1, 7, 16, 8);
      case 1792: return X32(1792, 16, 2, false, 1, 7, 16, 16);
      case 3584: return X32(3584, 16, 1, false, 1, 7, 16, 16, 2);
    }
  } else {
    switch (d.n) {
      case 240: return X32(240, 16, 32, true, 1, 15, 16);
      case 480: return X32(480, 32, 32, true, 1, 15, 16, 2);
      case 960: return X32(960, 32, 32, true, 4, 15, 16, 4);
      case 1920: return X32(1920, 32, 16, true, 4, 15, 16, 8);
      case 3840: return X32(3840, 32, 8, true, 4, 15, 16, 16);
      case 720: return X32(720, 16, 16, true, 4, 15, 3, 16);
      case 1440: return X32(1440, 32, 16, true, 4, 15, 3, 16, 2);
      case 2880: return X32(2880, 32, 8, true, 4, 15, 3, 16, 4);
      case 1200: return X32(1200, 16, 8, true, 4, 15, 5, 16);
      case 2400: return X32(2400, 32, 8, true, 4, 15, 5, 16, 2);
      case 112: return X32(112, 16, 32, true, 1, 7, 16);
      case 224: return X32(224, 32, 32, true, 1, 7, 16, 2);
      case 448: return X32(448, 32, 32, true, 1, 7, 16, 4);
      case 896: return X32(896, 32, 32, true, 4, 7, 16, 8);
      case 1792: return X32(1792, 32, 16, true, 4, 7, 16, 16);
      case 3584: return X32(3584, 32, 8, true, 4, 7, 16, 16, 2);
    }
  }
  return hipErrorInvalidValue;
}

// packed-real rows of 2 n reals (MODE_R2C_H / MODE_C2R_H, fft_real_f64.hip) on the same row plans: the Hermitian pass runs in
// the geometry of the side it sits on (after the last stage for r2c, before the first for c2r).  Plain rows only.
template <int MODE>
static hipError_t halfv_f32(const PassDesc &d, const void *in, void *out, hipStream_t s) {
  if (d.tr_dir || d.ub_p > 1) return hipErrorInvalidValue;
  switch (d.n) {
    case 240: return launch_pow2_one<float, 240, 16, 16, false, true, 1, 0, MODE, false, 15, 16>(d, in, out, s);
    case 480: return launch_pow2_one<float, 480, 16, 8, false, true, 1, 0, MODE, false, 15, 16, 2>(d, in, out, s);
    case 960: return launch_pow2_one<float, 960, 16, 4, false, true, 1, 0, MODE, false, 15, 16, 4>(d, in, out, s);
    case 1920: return launch_pow2_one<float, 1920, 16, 2, false, true, 1, 0, MODE, false, 15, 16, 8>(d, in, out, s);
    case 3840: return launch_pow2_one<float, 3840, 16, 1, false, true, 1, 0, MODE, false, 15, 16, 16>(d, in, out, s);
    case 720: return launch_pow2_one<float, 720, 16, 8, false, true, 1, 0, MODE, false, 15, 3, 16>(d, in, out, s);
    case 1440: return launch_pow2_one<float, 1440, 16, 4, false, true, 1, 0, MODE, false, 15, 3, 16, 2>(d, in, out, s);
    case 2880: return launch_pow2_one<float, 2880, 16, 2, false, true, 1, 0, MODE, false, 15, 3, 16, 4>(d, in, out, s);
    case 1200: return launch_pow2_one<float, 1200, 16, 4, false, true, 1, 0, MODE, false, 15, 5, 16>(d, in, out, s);
    case 2400: return launch_pow2_one<float, 2400, 16, 2, false, true, 1, 0, MODE, false, 15, 5, 16, 2>(d, in, out, s);
    case 112: return launch_pow2_one<float, 112, 16, 16, false, true, 1, 0, MODE, false, 7, 16>(d, in, out, s);
    case 224: return launch_pow2_one<float, 224, 16, 16, false, true, 1, 0, MODE, false, 7, 16, 2>(d, in, out, s);
    case 448: return launch_pow2_one<float, 448, 16, 8, false, true, 1, 0, MODE, false, 7, 16, 4>(d, in, out, s);
    case 896: return launch_pow2_one<float, 896, 16, 4, false, true, 1, 0, MODE, false, 7, 16, 8>(d, in, out, s);
    case 1792: return launch_pow2_one<float, 1792, 16, 2, false, true, 1, 0, MODE, false, 7, 16, 16>(d, in, out, s);
    case 3584: return launch_pow2_one<float, 3584, 16, 1, false, true, 1, 0, MODE, false, 7, 16, 16, 2>(d, in, out, s);
  }
  return hipErrorInvalidValue;
}

hipError_t launch_real_half_mixv_f32(const PassDesc &d, const void *in, void *out, hipStream_t s) {
  if (d.mode == MODE_R2C_H) return halfv_f32<MODE_R2C_H>(d, in, out, s);
  if (d.mode == MODE_C2R_H) return halfv_f32<MODE_C2R_H>(d, in, out, s);
  return hipErrorInvalidValue;
}

}  // namespace gfft
