// fp32 (complex64) instantiations of the power-of-two pass kernels (the `fftwf_*` clone of the
// reference, setup.py:93-111).  A complex64 is 8 bytes, so R = 16 costs the registers R = 8
// costs in fp64, and 16 adjacent columns make the 128-byte segment.
#include "fft_pow2_impl.h"

namespace gfft {

#define P32(N, R, T, COLS, SPLIT, MINW, ...) \
  launch_pow2_inst<float, N, R, T, COLS, SPLIT, MINW, 0, __VA_ARGS__>(d, in, out, s)

bool pow2_supported_f32(int n) { return n >= 16 && n <= 4096 && (n & (n - 1)) == 0; }

hipError_t launch_pow2_f32(const PassDesc &d, bool cols, int variant, const void *in, void *out,
                           hipStream_t s) {
  (void)variant;
  if (!cols) {
    switch (d.n) {
      case 16: return P32(16, 4, 16, false, false, 1, 4, 4);
      case 32: return P32(32, 8, 16, false, false, 1, 8, 4);
      case 64: return P32(64, 8, 8, false, false, 1, 8, 8);
      case 128: return P32(128, 8, 4, false, false, 1, 8, 8, 2);
      case 256: return P32(256, 16, 4, false, false, 1, 16, 16);
      case 512: return P32(512, 8, 1, false, false, 1, 8, 8, 8);
      case 1024: return P32(1024, 16, 1, false, false, 1, 16, 16, 4);
      case 2048: return P32(2048, 16, 1, false, false, 1, 16, 16, 8);
      case 4096: return P32(4096, 16, 1, false, false, 1, 16, 16, 16);
    }
  } else if (d.mode != MODE_C2C || d.tw_hi || d.tr_dir || d.out_es == 1 || d.in_es == 1) {
    // real modes along a strided axis and the four-step passes: lean R = 8 plans
    switch (d.n) {
      case 16: return P32(16, 4, 16, true, false, 1, 4, 4);
      case 32: return P32(32, 8, 16, true, false, 1, 8, 4);
      case 64: return P32(64, 8, 16, true, false, 1, 8, 8);
      case 128: return P32(128, 8, 16, true, false, 1, 8, 8, 2);
      case 256: return P32(256, 8, 16, true, false, 1, 8, 8, 4);
      case 512: return P32(512, 8, 16, true, true, 1, 8, 8, 8);
      case 1024: return P32(1024, 8, 8, true, false, 1, 8, 8, 8, 2);
      case 2048: return P32(2048, 8, 4, true, false, 1, 8, 8, 8, 4);
      case 4096: return P32(4096, 8, 2, true, false, 1, 8, 8, 8, 8);
    }
  } else {
    switch (d.n) {
      case 16: return P32(16, 4, 16, true, false, 1, 4, 4);
      case 32: return P32(32, 8, 16, true, false, 1, 8, 4);
      case 64: return P32(64, 8, 16, true, false, 1, 8, 8);
      case 128: return P32(128, 8, 16, true, false, 1, 8, 8, 2);
      case 256: return P32(256, 16, 16, true, false, 1, 16, 16);
      case 512: return P32(512, 8, 16, true, true, 1, 8, 8, 8);
      case 1024: return P32(1024, 16, 16, true, true, 1, 16, 16, 4);
      case 2048: return P32(2048, 16, 8, true, true, 4, 16, 16, 8);
      case 4096: return P32(4096, 16, 4, true, true, 4, 16, 16, 16);
    }
  }
  return hipErrorInvalidValue;
}

}  // namespace gfft
