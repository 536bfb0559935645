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

// Round 5: the other kinds on the same two tile shapes -- 32 adjacent columns on 32 values per thread (256-byte hand-off
// segments), 16 rows with a row inside one wave --: the four-step pairs of a length-2^20 transform (config C2 in
// complex64: 128 x 2^20 ran as two launches at 0.245 of the 2 S roofline) and the batched 2-D pair on contiguous planes.
struct Fused1024F32 {
  typedef PassCfg<float, 1024, 16, 16, false, false, 1 | 2048 | 8192, MODE_C2C, false, 16, 16, 4> RowsToRing;
  typedef RowsFromRingF32 RowsFromRing;
  typedef ColsToRingF32 ColsToRing;
  typedef PassCfg<float, 1024, 32, 32, true, true, 2 | 8 | 4096 | 8192, MODE_C2C, false, 16, 16, 4> ColsFromRing;
  typedef PassCfg<float, 1024, 32, 32, true, true, 1 | 32 | 2048 | 8192, MODE_C2C, true, 16, 16, 4> FourStepFirst;
  typedef PassCfg<float, 1024, 32, 32, true, true, 1 | 2048 | 8192, MODE_C2C, true, 16, 16, 4> FourStepFirstNat;
  typedef PassCfg<float, 1024, 16, 16, false, true, 2 | 32 | 4096 | 8192, MODE_C2C, false, 16, 16, 4> RowsFromRingT;      // (the transposing store works on split planes)
};

// the four-step first passes on the stand-alone table's shape instead -- 16 columns on 16 values per thread, 128-byte
// hand-off segments, no spills -- (option fuse2_f32 = 3; A/B)
struct Fused1024F32Lean : Fused1024F32 {
  typedef PassCfg<float, 1024, 16, 16, true, true, 1 | 32 | 2048 | 8192, MODE_C2C, true, 16, 16, 4> FourStepFirst;
  typedef PassCfg<float, 1024, 16, 16, true, true, 1 | 2048 | 8192, MODE_C2C, true, 16, 16, 4> FourStepFirstNat;
};
int g_fuse2_f32_lean = 0;

bool fused2_supported_f32(int kind, int n_a, int n_b) {
  if (n_a != 1024 || n_b != 1024) return false;
  return kind == FUSED_COLS_ROWS || kind == FUSED_FOURSTEP || kind == FUSED_FOURSTEP_ROWS || kind == FUSED_PLANES_2D;
}

int fused2_tiles_f32(int kind, const PassDesc &dA, const PassDesc &dB, int *ta, int *tb) {
  if (g_fuse2_f32_lean) return fused2_tiles_kind<Fused1024F32Lean>(kind, dA, dB, ta, tb);
  return fused2_tiles_kind<Fused1024F32>(kind, dA, dB, ta, tb);
}

hipError_t launch_fused2_f32(int kind, const PassDesc &dA, const PassDesc &dB, const PassDesc *dev, const FusedDesc &f,
                             const void *in, void *ring, void *out, hipStream_t s) {
  if (kind == FUSED_ROWS_COLS) return hipErrorInvalidValue;          // (not built: the forward order the 3-D schedule does not take)
  if (g_fuse2_f32_lean && (kind == FUSED_FOURSTEP || kind == FUSED_FOURSTEP_ROWS)) return launch_fused2_kind<Fused1024F32Lean>(kind, dA, dB, dev, f, in, ring, out, s);
  return launch_fused2_kind<Fused1024F32>(kind, dA, dB, dev, f, in, ring, out, s);
}

}  // namespace gfft
