// fp64 (complex128) one-pass kernels for lengths 3 x 5 x 2^k (and 3^2 x 5 x 2^k, 3 x 5^2 x 2^k: 240 ... 3840) and 7 x 2^k
// (112 ... 3584) -- the grid sizes
// between the powers of two that are neither 3-smooth nor 5-smooth times a power of two (960, 1920, 720, 1200 ...), which
// rounds 1-4 ran as TWO passes per axis (960 = 48 x 20: 960^3 complex128 at 0.29 of the 2 S roofline).
// No single number of values per thread serves a radix-15 and a radix-16 stage; here every stage keeps as many as its
// radix divides -- 15 of the 16 in the radix-15 stage -- on a column of max_s n / R_s threads (Geo / StageV,
// fft_pow2_impl.h).  The radix-15 stage comes FIRST: its scatter runs in odd multiples (no LDS slot padding needed) and the
// load side then uses every thread of the workgroup.  Plain complex passes only (TABLE_FLAGS 8): natural layouts, no fused
// truncation, no four-step twiddle -- the planner keeps other uses of these lengths on the two-pass / generic paths.
// The reference's own tests live on such sizes (tests/test_libfft.py:26-27, tests/test_mpifft.py:57-111).
#include "fft_pow2_impl.h"

namespace gfft {

#define X64(N, R, T, COLS, MINW, ...) \
  launch_pow2_inst<double, N, R, T, COLS, true, MINW, 8, __VA_ARGS__>(d, in, out, s)

bool mixv_supported(int n) {
  switch (n) {
    case 240: case 480: case 960: case 1920: case 3840:
    case 720: case 1440: case 2880:
    case 1200: case 2400:
    case 112: case 224: case 448: case 896: case 1792: case 3584:
      return true;
  }
  return false;
}

hipError_t launch_mixv_f64(const PassDesc &d, bool cols, const void *in, void *out, hipStream_t s) {
  if (d.mode != MODE_C2C) return hipErrorInvalidValue;
  if (!cols) {
    switch (d.n) {      // rows: whole rows per workgroup, >= 256 threads
      case 240: return X64(240, 16, 16, false, 1, 15, 16);
      case 480: return X64(480, 16, 8, false, 1, 15, 16, 2);
      case 960: return X64(960, 16, 4, false, 1, 15, 16, 4);
      case 1920: return X64(1920, 16, 2, false, 1, 15, 16, 8);
      case 3840: return X64(3840, 16, 1, false, 1, 15, 16, 16);
      case 720: return X64(720, 16, 8, false, 1, 15, 3, 16);
      case 1440: return X64(1440, 16, 4, false, 1, 15, 3, 16, 2);
      case 2880: return X64(2880, 16, 2, false, 1, 15, 3, 16, 4);
      case 1200: return X64(1200, 16, 4, false, 1, 15, 5, 16);
      case 2400: return X64(2400, 16, 2, false, 1, 15, 5, 16, 2);
      // 7 x 2^k: the radix-7 stage keeps 14 of the 16 values
      case 112: return X64(112, 16, 16, false, 1, 7, 16);
      case 224: return X64(224, 16, 16, false, 1, 7, 16, 2);
      case 448: return X64(448, 16, 8, false, 1, 7, 16, 4);
      case 896: return X64(896, 16, 4, false, 1, 7, 16, 8);
      case 1792: return X64(1792, 16, 2, false, 1, 7, 16, 16);
      case 3584: return X64(3584, 16, 1, false, 1, 7, 16, 16, 2);
    }
  } else {
    switch (d.n) {      // strided: 16 adjacent columns = 256-byte segments while the tile fits 1024 threads and the LDS
      case 240: return X64(240, 16, 16, true, 1, 15, 16);
      case 480: return X64(480, 16, 16, true, 1, 15, 16, 2);
      case 960: return X64(960, 16, 16, true, 4, 15, 16, 4);
      case 1920: return X64(1920, 16, 8, true, 4, 15, 16, 8);
      case 3840: return X64(3840, 16, 4, true, 4, 15, 16, 16);
      case 720: return X64(720, 16, 16, true, 4, 15, 3, 16);
      case 1440: return X64(1440, 16, 8, true, 4, 15, 3, 16, 2);
      case 2880: return X64(2880, 16, 4, true, 4, 15, 3, 16, 4);
      case 1200: return X64(1200, 16, 8, true, 4, 15, 5, 16);
      case 2400: return X64(2400, 16, 4, true, 4, 15, 5, 16, 2);
      case 112: return X64(112, 16, 16, true, 1, 7, 16);
      case 224: return X64(224, 16, 16, true, 1, 7, 16, 2);
      case 448: return X64(448, 16, 16, true, 1, 7, 16, 4);
      case 896: return X64(896, 16, 16, true, 4, 7, 16, 8);
      case 1792: return X64(1792, 16, 8, true, 4, 7, 16, 16);
      case 3584: return X64(3584, 16, 4, true, 4, 7, 16, 16, 2);
    }
  }
  return hipErrorInvalidValue;
}

// packed-real rows of 2 n reals (MODE_R2C_H / MODE_C2R_H, fft_real_f64.hip) on the same row plans: the Hermitian pass runs in
// the geometry of the side it sits on (after the last stage for r2c, before the first for c2r).  Plain rows only.
template <int MODE>
static hipError_t halfv_f64(const PassDesc &d, const void *in, void *out, hipStream_t s) {
  if (d.tr_dir || d.ub_p > 1) return hipErrorInvalidValue;
  switch (d.n) {
    case 240: return launch_pow2_one<double, 240, 16, 16, false, true, 1, 0, MODE, false, 15, 16>(d, in, out, s);
    case 480: return launch_pow2_one<double, 480, 16, 8, false, true, 1, 0, MODE, false, 15, 16, 2>(d, in, out, s);
    case 960: return launch_pow2_one<double, 960, 16, 4, false, true, 1, 0, MODE, false, 15, 16, 4>(d, in, out, s);
    case 1920: return launch_pow2_one<double, 1920, 16, 2, false, true, 1, 0, MODE, false, 15, 16, 8>(d, in, out, s);
    case 3840: return launch_pow2_one<double, 3840, 16, 1, false, true, 1, 0, MODE, false, 15, 16, 16>(d, in, out, s);
    case 720: return launch_pow2_one<double, 720, 16, 8, false, true, 1, 0, MODE, false, 15, 3, 16>(d, in, out, s);
    case 1440: return launch_pow2_one<double, 1440, 16, 4, false, true, 1, 0, MODE, false, 15, 3, 16, 2>(d, in, out, s);
    case 2880: return launch_pow2_one<double, 2880, 16, 2, false, true, 1, 0, MODE, false, 15, 3, 16, 4>(d, in, out, s);
    case 1200: return launch_pow2_one<double, 1200, 16, 4, false, true, 1, 0, MODE, false, 15, 5, 16>(d, in, out, s);
    case 2400: return launch_pow2_one<double, 2400, 16, 2, false, true, 1, 0, MODE, false, 15, 5, 16, 2>(d, in, out, s);
    case 112: return launch_pow2_one<double, 112, 16, 16, false, true, 1, 0, MODE, false, 7, 16>(d, in, out, s);
    case 224: return launch_pow2_one<double, 224, 16, 16, false, true, 1, 0, MODE, false, 7, 16, 2>(d, in, out, s);
    case 448: return launch_pow2_one<double, 448, 16, 8, false, true, 1, 0, MODE, false, 7, 16, 4>(d, in, out, s);
    case 896: return launch_pow2_one<double, 896, 16, 4, false, true, 1, 0, MODE, false, 7, 16, 8>(d, in, out, s);
    case 1792: return launch_pow2_one<double, 1792, 16, 2, false, true, 1, 0, MODE, false, 7, 16, 16>(d, in, out, s);
    case 3584: return launch_pow2_one<double, 3584, 16, 1, false, true, 1, 0, MODE, false, 7, 16, 16, 2>(d, in, out, s);
  }
  return hipErrorInvalidValue;
}

hipError_t launch_real_half_mixv_f64(const PassDesc &d, const void *in, void *out, hipStream_t s) {
  if (d.mode == MODE_R2C_H) return halfv_f64<MODE_R2C_H>(d, in, out, s);
  if (d.mode == MODE_C2R_H) return halfv_f64<MODE_C2R_H>(d, in, out, s);
  return hipErrorInvalidValue;
}

}  // namespace gfft
