// fp32 (complex64) fused pass pairs.  Round 3 found the complex64 pair slower than its two stand-alone passes (1024^3:
// 22.1 -> 25.7 ms per step) and left it out; re-measured in round 4 -- the row tiles now exchange inside their waves --
// it still loses with the ring of 12 planes / 6 ahead that complex128 wants (21.1 -> 21.9 ms) and GAINS with the same
// lead in bytes, 24 / 12 planes of 8 MiB: 20.94 -> 18.88 ms per step, the pair 6.04 ms against 3.34 + 3.53
// (profiles/r04_ab_fuse2_f32.txt; make_fused2 sizes the ring in bytes since).  Kinds: [strided n -> rows n] of the
// complex 3-D schedule, n = 1024.
#include "fft_fused_impl.h"

namespace gfft {

//                          real   N     R   T   COLS   SPLIT  FLAGS                  MODE      BIGTW  radices
typedef PassCfg<float, 1024, 32, 32, true, true, 1 | 8 | 2048 | 8192, MODE_C2C, false, 16, 16, 4> ColsToRingF32;      // 32 columns = 256-byte segments
typedef PassCfg<float, 1024, 16, 16, false, false, 2 | 4096 | 8192, MODE_C2C, false, 16, 16, 4> RowsFromRingF32;    // a row per wave

bool fused2_supported_f32(int kind, int n_a, int n_b) { return kind == FUSED_COLS_ROWS && n_a == 1024 && n_b == 1024; }

int fused2_tiles_f32(int kind, const PassDesc &dA, const PassDesc &dB, int *ta, int *tb) {
  if (kind != FUSED_COLS_ROWS) return -1;
  *ta = (int)ColsToRingF32::ntiles(dA);
  *tb = (int)RowsFromRingF32::ntiles(dB);
  return 0;
}

hipError_t launch_fused2_f32(int kind, const PassDesc &dA, const PassDesc &dB, const PassDesc *dev, const FusedDesc &f,
                             const void *in, void *ring, void *out, hipStream_t s) {
  if (kind != FUSED_COLS_ROWS) return hipErrorInvalidValue;
  return launch_fused2<ColsToRingF32, RowsFromRingF32>(dA, dB, dev, f, in, ring, out, s);
}

}  // namespace gfft
