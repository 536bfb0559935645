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
// segments), 16 rows with a row inside one wave --: the four-step pair of a length-2^20 transform and the batched 2-D pair
// on contiguous planes.  Same arrays, plans alternating (tools/serial_ab_probe.py, profiles/r05_c2_pairs.txt):
//   128 x 2^20 complex64 (config C2's shape in fp32): two launches 1.097 ms -> [strided -> strided] pair 0.979 ms (-10.8 %);
//     the [strided -> rows, transposed on store] form that complex128 prefers: 1.009 ms (its first pass spills 128 bytes
//     at 32 values per thread + four-step twiddle in 128 VGPRs) -- not built into the product;
//   (256,1024,1024) complex64 over axes (1, 2): 1.817 -> 1.567 ms (-13.7 %).
// Half the bytes per butterfly of complex128: the LDS exchanges (4-byte planes, twice the DS instructions per byte) and the
// butterflies weigh twice as much in a tile, and one resident workgroup per CU cannot overlap them with its memory phases.
struct Fused1024F32 {
  typedef PassCfg<float, 1024, 16, 16, false, false, 1 | 2048 | 8192, MODE_C2C, false, 16, 16, 4> RowsToRing;
  typedef PassCfg<float, 1024, 32, 32, true, true, 2 | 8 | 4096 | 8192, MODE_C2C, false, 16, 16, 4> ColsFromRing;
  typedef PassCfg<float, 1024, 32, 32, true, true, 1 | 32 | 2048 | 8192, MODE_C2C, true, 16, 16, 4> FourStepFirst;
  // the strided side of the slab pairs: the array side is an all-to-all buffer of equal blocks (FLAGS 32768 input / 65536 output, FUSED_PLANES_2D_B / _CR_B)
  typedef PassCfg<float, 1024, 32, 32, true, true, 1 | 8 | 2048 | 8192 | 32768, MODE_C2C, false, 16, 16, 4> ColsToRingB;
  typedef PassCfg<float, 1024, 32, 32, true, true, 2 | 8 | 4096 | 8192 | 65536, MODE_C2C, false, 16, 16, 4> ColsFromRingB;
};

// Round 6: n = 512, [strided -> rows], both tiles on 32 values per thread / radices 32 x 16 = ONE exchange, 512 threads of up to 256 VGPRs (167, no
// spills): 32 columns (256-byte hand-off segments) and 32 rows (16 lanes per row: the exchange stays inside the wave) -- 128 KiB either tile.
// Plans alternating on the same arrays (tools/unequal_pair_probe.py f32n512, profiles/r06_f32_n512_pair_probe.txt): slab pairs (256,512,512)
// 0.397 / 0.392 -> 0.359 / 0.368 ms, (1024,512,512) over 8 blocks 1.743 / 1.660 -> 1.374 / 1.424 ms (0.62 -> 0.78 of 8 TB/s); 512^3 c64 per
// fwd + bwd step 2.436 -> 2.172 ms.  (First form, not kept: 8 rows per tile with a row inside one wave on 8 values per thread -- 32 KiB row
// tiles, four times the tickets: the pair LOST 5-26 %.)  Option fuse2_f32_n512.
struct Fused512F32 {
  typedef PassCfg<float, 512, 32, 32, true, true, 1 | 8 | 2048 | 8192, MODE_C2C, false, 32, 16> ColsToRing;
  typedef PassCfg<float, 512, 32, 32, true, true, 1 | 8 | 2048 | 8192 | 32768, MODE_C2C, false, 32, 16> ColsToRingB;
  typedef PassCfg<float, 512, 32, 32, false, false, 2 | 4096 | 8192, MODE_C2C, false, 32, 16> RowsFromRing;
};
int g_fuse2_f32_n512 = 1;

bool fused2_supported_f32(int kind, int n_a, int n_b) {
  if (n_a == 512 && n_b == 512) return g_fuse2_f32_n512 != 0 && (kind == FUSED_COLS_ROWS || kind == FUSED_PLANES_CR_B);
  if (n_a != 1024 || n_b != 1024) return false;
  return kind == FUSED_COLS_ROWS || kind == FUSED_FOURSTEP || kind == FUSED_PLANES_2D || kind == FUSED_PLANES_2D_B || kind == FUSED_PLANES_CR_B;
}

int fused2_tiles_f32(int kind, const PassDesc &dA, const PassDesc &dB, int *ta, int *tb) {
  if (dA.n == 512) {
    if (kind != FUSED_COLS_ROWS && kind != FUSED_PLANES_CR_B) return -1;
    *ta = (int)Fused512F32::ColsToRing::ntiles(dA);
    *tb = (int)Fused512F32::RowsFromRing::ntiles(dB);
    return 0;
  }
  switch (kind) {
    case FUSED_COLS_ROWS: *ta = (int)ColsToRingF32::ntiles(dA); *tb = (int)RowsFromRingF32::ntiles(dB); return 0;
    case FUSED_FOURSTEP: *ta = (int)Fused1024F32::FourStepFirst::ntiles(dA); *tb = (int)Fused1024F32::ColsFromRing::ntiles(dB); return 0;
    case FUSED_PLANES_2D_B:
    case FUSED_PLANES_2D: *ta = (int)Fused1024F32::RowsToRing::ntiles(dA); *tb = (int)Fused1024F32::ColsFromRing::ntiles(dB); return 0;
    case FUSED_PLANES_CR_B: *ta = (int)Fused1024F32::ColsToRingB::ntiles(dA); *tb = (int)RowsFromRingF32::ntiles(dB); return 0;
  }
  return -1;
}

hipError_t launch_fused2_f32(int kind, const PassDesc &dA, const PassDesc &dB, const PassDesc *dev, const FusedDesc &f,
                             const void *in, void *ring, void *out, hipStream_t s) {
  if (dA.n == 512) {
    if (kind == FUSED_COLS_ROWS) return launch_fused2<Fused512F32::ColsToRing, Fused512F32::RowsFromRing>(dA, dB, dev, f, in, ring, out, s);
    if (kind == FUSED_PLANES_CR_B) return launch_fused2<Fused512F32::ColsToRingB, Fused512F32::RowsFromRing>(dA, dB, dev, f, in, ring, out, s);
    return hipErrorInvalidValue;
  }
  switch (kind) {
    case FUSED_COLS_ROWS: return launch_fused2<ColsToRingF32, RowsFromRingF32>(dA, dB, dev, f, in, ring, out, s);
    case FUSED_FOURSTEP: return launch_fused2<Fused1024F32::FourStepFirst, Fused1024F32::ColsFromRing>(dA, dB, dev, f, in, ring, out, s);
    case FUSED_PLANES_2D: return launch_fused2<Fused1024F32::RowsToRing, Fused1024F32::ColsFromRing>(dA, dB, dev, f, in, ring, out, s);
    case FUSED_PLANES_2D_B: return launch_fused2<Fused1024F32::RowsToRing, Fused1024F32::ColsFromRingB>(dA, dB, dev, f, in, ring, out, s);
    case FUSED_PLANES_CR_B: return launch_fused2<Fused1024F32::ColsToRingB, RowsFromRingF32>(dA, dB, dev, f, in, ring, out, s);
  }
  return hipErrorInvalidValue;
}

}  // namespace gfft
