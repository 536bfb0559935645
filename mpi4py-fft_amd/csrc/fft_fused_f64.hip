// fp64 fused pass pairs: two dependent passes of a transform in one persistent launch, the intermediate
// handed over through a ring buffer that lives in the Infinity Cache (fft_pow2_impl.h, fft_fused2_kernel).
//   [rows n] -> [strided n]   passes 1 + 2 of a complex 3-D schedule, forward order (plan.cpp plan_fused3; built, not the default)
//   [strided n] -> [rows n]   passes 2 + 3 of the complex 3-D schedule as both directions run it
//   four-step n x n           both passes of a length-n^2 transform (n = 1024: BASELINE config C2), second pass strided or rows
// Every pass works on 16 lines per workgroup -- the row pass therefore takes 16 rows per tile instead of the stand-alone
// table's 4 -- with the hand-off side at system scope.
#include "fft_fused_impl.h"

namespace gfft {

//                                  real   N     R   T   COLS   SPLIT FLAGS                 MODE      BIGTW  radices
// (FLAGS 1 / 2: the streams that are NOT the hand-off -- A's loads, B's stores -- are non-temporal, so that
// they do not push the ring out of the Infinity Cache: 1024^3 per step 37.35 -> 36.26 ms, clean A/B)
// Measured and NOT kept (tools/ab_option_probe.py fuse2 0,1 <dtype> <n>, fwd + bwd per step): fp64 n = 512
// (512^3: 5.03 ms unfused, 7.01 fused; four-step 2^18: 0.90 -> 1.32 ms) -- a 128 KiB tile is over in ~10 us, so
// the per-ticket costs (ticket, counters, write-through acknowledgements) weigh twice as much --, and fp32
// n = 1024 (1024^3 complex64: 22.1 ms unfused, 25.7 fused; four-step level) -- half the bytes per butterfly,
// so the butterfly / LDS phases, which a single resident workgroup per CU cannot overlap with its memory
// phases, dominate the tile.  The pairs pay where the tile is memory heavy: fp64 at n = 1024.
#ifdef GFFT_VARIANTS
// variant 2: 8 lines per tile, 512 threads, two workgroups per CU -- built to fill the gaps one workgroup per
// CU leaves between load, butterflies and store; measured slower (21.0 / 20.4 ms against 18.5 / 19.1 per
// 1024^3 direction: its hand-off accesses are 128-byte pieces at system scope)
struct Fused1024x8 {
  typedef PassCfg<double, 1024, 16, 8, false, true, 1 | 2048 | 8192, MODE_C2C, false, 16, 16, 4> RowsToRing;
  typedef PassCfg<double, 1024, 16, 8, false, true, 2 | 4096 | 8192, MODE_C2C, false, 16, 16, 4> RowsFromRing;
  typedef PassCfg<double, 1024, 16, 8, true, true, 1 | 8 | 2048 | 8192, MODE_C2C, false, 16, 16, 4> ColsToRing;
  typedef PassCfg<double, 1024, 16, 8, true, true, 2 | 8 | 4096 | 8192, MODE_C2C, false, 16, 16, 4> ColsFromRing;
  typedef PassCfg<double, 1024, 16, 8, true, true, 1 | 32 | 2048 | 8192, MODE_C2C, true, 16, 16, 4> FourStepFirst;
  typedef PassCfg<double, 1024, 16, 8, true, true, 1 | 2048 | 8192, MODE_C2C, true, 16, 16, 4> FourStepFirstNat;
  typedef PassCfg<double, 1024, 16, 8, false, true, 2 | 32 | 4096 | 8192, MODE_C2C, false, 16, 16, 4> RowsFromRingT;
};
// variant 4: the two ideas together -- 8 lines per tile AND one exchange (32 values per thread): 256-thread workgroups of
// up to 256 VGPRs, two per CU
struct Fused1024x8R32 {
  typedef PassCfg<double, 1024, 32, 8, false, true, 1 | 2048 | 8192, MODE_C2C, false, 32, 32> RowsToRing;
  typedef PassCfg<double, 1024, 32, 8, false, true, 2 | 4096 | 8192, MODE_C2C, false, 32, 32> RowsFromRing;
  typedef PassCfg<double, 1024, 32, 8, true, true, 1 | 8 | 2048 | 8192, MODE_C2C, false, 32, 32> ColsToRing;
  typedef PassCfg<double, 1024, 32, 8, true, true, 2 | 8 | 4096 | 8192, MODE_C2C, false, 32, 32> ColsFromRing;
  typedef PassCfg<double, 1024, 32, 8, true, true, 1 | 32 | 2048 | 8192, MODE_C2C, true, 32, 32> FourStepFirst;
  typedef PassCfg<double, 1024, 32, 8, true, true, 1 | 2048 | 8192, MODE_C2C, true, 32, 32> FourStepFirstNat;
  typedef PassCfg<double, 1024, 32, 8, false, true, 2 | 32 | 4096 | 8192, MODE_C2C, false, 32, 32> RowsFromRingT;
};
#endif

// The default since round 4: 32 values per thread, radices 32 x 32 -- ONE LDS exchange per tile instead of two (4
// barriers on the strided tiles instead of 8), on 512-thread workgroups (8 waves, up to 256 VGPRs each); same 16 lines
// per tile, same 256-byte hand-off segments
struct Fused1024R32 {
  typedef PassCfg<double, 1024, 32, 16, false, true, 1 | 2048 | 8192, MODE_C2C, false, 32, 32> RowsToRing;
  typedef PassCfg<double, 1024, 32, 16, false, true, 2 | 4096 | 8192, MODE_C2C, false, 32, 32> RowsFromRing;
  typedef PassCfg<double, 1024, 32, 16, true, true, 1 | 8 | 2048 | 8192, MODE_C2C, false, 32, 32> ColsToRing;
  typedef PassCfg<double, 1024, 32, 16, true, true, 2 | 8 | 4096 | 8192, MODE_C2C, false, 32, 32> ColsFromRing;
  typedef PassCfg<double, 1024, 32, 16, true, true, 1 | 32 | 2048 | 8192, MODE_C2C, true, 32, 32> FourStepFirst;
  typedef PassCfg<double, 1024, 32, 16, true, true, 1 | 2048 | 8192, MODE_C2C, true, 32, 32> FourStepFirstNat;
  typedef PassCfg<double, 1024, 32, 16, false, true, 2 | 32 | 4096 | 8192, MODE_C2C, false, 32, 32> RowsFromRingT;
  // the strided side of the slab pairs: the array side is an all-to-all buffer of equal blocks (FLAGS 32768 input / 65536 output, FUSED_PLANES_2D_B / _CR_B)
  typedef PassCfg<double, 1024, 32, 16, true, true, 1 | 8 | 2048 | 8192 | 32768, MODE_C2C, false, 32, 32> ColsToRingB;
  typedef PassCfg<double, 1024, 32, 16, true, true, 2 | 8 | 4096 | 8192 | 65536, MODE_C2C, false, 32, 32> ColsFromRingB;
};

// n = 512: 32 values per thread, radices 32 x 16, 256-thread workgroups on 16 lines -- 128 KiB tiles, TWO workgroups per CU
// (round 3 measured the n = 512 pairs on 1024-thread workgroups of 8 values per thread and found them 40 % slower than
// their stand-alone passes)
struct Fused512R32 {
  typedef PassCfg<double, 512, 32, 16, false, true, 1 | 2048 | 8192, MODE_C2C, false, 32, 16> RowsToRing;
  typedef PassCfg<double, 512, 32, 16, false, true, 2 | 4096 | 8192, MODE_C2C, false, 32, 16> RowsFromRing;
  typedef PassCfg<double, 512, 32, 16, true, true, 1 | 8 | 2048 | 8192, MODE_C2C, false, 32, 16> ColsToRing;
  typedef PassCfg<double, 512, 32, 16, true, true, 2 | 8 | 4096 | 8192, MODE_C2C, false, 32, 16> ColsFromRing;
  typedef PassCfg<double, 512, 32, 16, true, true, 1 | 32 | 2048 | 8192, MODE_C2C, true, 32, 16> FourStepFirst;
  typedef PassCfg<double, 512, 32, 16, true, true, 1 | 2048 | 8192, MODE_C2C, true, 32, 16> FourStepFirstNat;
  typedef PassCfg<double, 512, 32, 16, false, true, 2 | 32 | 4096 | 8192, MODE_C2C, false, 32, 16> RowsFromRingT;
  typedef PassCfg<double, 512, 32, 16, true, true, 1 | 8 | 2048 | 8192 | 32768, MODE_C2C, false, 32, 16> ColsToRingB;
  typedef PassCfg<double, 512, 32, 16, true, true, 2 | 8 | 4096 | 8192 | 65536, MODE_C2C, false, 32, 16> ColsFromRingB;
};

// Unequal pairs (round 6): planes of 512 x 1024 or 1024 x 512 points -- non-cubic grids, e.g. (512,1024,1024) on one rank or as the
// slabs of a distributed transform.  Both passes share one 512-thread workgroup shape: the n = 1024 side is Fused1024R32's (16 lines
// per tile), the n = 512 side takes 32 lines per tile (512-byte hand-off segments, the same 256 KiB of values per tile).
struct Fused512T32 {
  typedef PassCfg<double, 512, 32, 32, false, true, 2 | 4096 | 8192, MODE_C2C, false, 32, 16> RowsFromRing;
  typedef PassCfg<double, 512, 32, 32, true, true, 1 | 8 | 2048 | 8192, MODE_C2C, false, 32, 16> ColsToRing;
  typedef PassCfg<double, 512, 32, 32, true, true, 1 | 8 | 2048 | 8192 | 32768, MODE_C2C, false, 32, 16> ColsToRingB;
};

// variant: 1 = the default, 32 values per thread / one exchange (Fused1024R32; the round-3 kernels -- 16 values per thread / two
// exchanges on 1024 threads -- were kept as variant 3 for A/B until round 6: 1024^3 per step 33.4 -> 32.6 ms, C2 0.707 -> 0.677 ms
// with variant 1, profiles/r04_ab_fuse2_variants.txt); 2 / 4 = (make VARIANTS=1) 8 lines per tile,
// two workgroups per CU, with 16 / 32 values per thread: 40.6 / 39.7 ms per step, a quarter SLOWER -- what bounds the fused
// launch is the traffic its CUs can move across the L2 boundary, and 128-byte pieces move less of it
// Round 5: the 3-D schedule's pair [strided n -> rows n] on two of the unequal-width stage lengths (fft_mixv_f64.hip): 32 values per
// thread on 512-thread workgroups of up to 256 VGPRs (219, no spills; the radix-15 / radix-7 stage keeps 30 / 28 of the 32), 16
// lines per tile.  Same arrays, plans alternating: 960^3 c128 per step 34.07 -> 30.54 ms with the pair on 16 values per thread / 1024
// threads (profiles/r05_ab_fuse2_mixv.txt), its launch 9.66 / 9.76 -> 9.58 / 9.62 ms on 32 values; 896^3 26.96 -> 26.03 ms, the launch
// 8.11 / 8.19 -> 7.80 / 7.83 ms (profiles/r05_ab_mixv_variants.txt).  (The STAND-ALONE strided kernels of these lengths lose 3-16 % on
// 32 values per thread, same file: they keep 16.)
struct Fused960 {
  typedef PassCfg<double, 960, 32, 16, true, true, 1 | 8 | 2048 | 8192, MODE_C2C, false, 15, 16, 4> ColsToRing;
  typedef PassCfg<double, 960, 32, 16, false, true, 2 | 4096 | 8192, MODE_C2C, false, 15, 16, 4> RowsFromRing;
};
struct Fused896 {
  typedef PassCfg<double, 896, 32, 16, true, true, 1 | 8 | 2048 | 8192, MODE_C2C, false, 7, 16, 8> ColsToRing;
  typedef PassCfg<double, 896, 32, 16, false, true, 2 | 4096 | 8192, MODE_C2C, false, 7, 16, 8> RowsFromRing;
};
int g_fuse2_mixv = 1;          // option fuse2_mixv

extern int g_fuse2_n512;
int g_fuse2_mixed = 1;        // option fuse2_mixed
bool fused2_supported_f64(int kind, int variant, int n_a, int n_b) {
  (void)kind;
  if (n_a != n_b)     // unequal planes: [strided -> rows] only (the 3-D schedule's pair, the slab pair of either direction)
    return g_fuse2_mixed != 0 && variant == 1 && ((n_a == 512 && n_b == 1024) || (n_a == 1024 && n_b == 512)) &&
           (kind == FUSED_COLS_ROWS || kind == FUSED_PLANES_CR_B);
  if (n_a == 960 || n_a == 896) return g_fuse2_mixv != 0 && variant == 1 && kind == FUSED_COLS_ROWS;
#ifdef GFFT_VARIANTS
  if (variant == 2 || variant == 4) return n_a == 1024;
#endif
  // (n = 512: measured for the 3-D schedule's pair only -- 512^3 per step 4.82 -> 4.27 ms with 24 planes of 4 MiB ahead,
  // 5.34 ms with 16: profiles/r04_ab_fuse2_n512.txt)
  // (... and the batched 2-D kind, [rows -> strided] on contiguous planes: (256,512,512) axes (1,2) 0.78 -> 0.69 ms, (512,512,512) 1.57 -> 1.34 ms)
  if (variant == 5) return n_a == 512 && (kind == FUSED_COLS_ROWS || kind == FUSED_PLANES_CR_B);      // the square n = 512 pair on 32 lines per tile (Fused512T32)
  if (variant == 1 && n_a == 512) return g_fuse2_n512 != 0 && (kind == FUSED_COLS_ROWS || kind == FUSED_PLANES_2D || kind == FUSED_PLANES_2D_B || kind == FUSED_PLANES_CR_B);
  if (kind == FUSED_PLANES_2D_B || kind == FUSED_PLANES_CR_B) return variant == 1 && n_a == 1024;
  return variant == 1 && n_a == 1024;
}
// option fuse2_n512: 0 = no n = 512 pairs, 1 = all of them on Fused512R32 (16 lines per tile, 256 threads, two workgroups per CU), 2 (the default since
// round 6) = the [strided -> rows] pairs on Fused512T32 instead: 32 lines per tile = 512-byte hand-off segments on 512 threads.  Plans alternating
// on the same arrays (tools/unequal_pair_probe.py n512, profiles/r06_n512_t32_probe.txt): the slab pair (256,512,512) -- config C3's local stages --
// 0.683 / 0.702 -> 0.608 / 0.612 ms (0.79 -> 0.88 of 8 TB/s), (512,512,512) over 4 blocks 1.317 / 1.358 -> 1.182 / 1.193 ms; the one-rank 512^3
// transform level (2.064 / 2.096 -> 2.085 / 2.089 ms).
int g_fuse2_n512 = 2;

int fused2_tiles_f64(int kind, int variant, const PassDesc &dA, const PassDesc &dB, int *tiles_a, int *tiles_b) {
#ifdef GFFT_VARIANTS
  if (variant == 2) return fused2_tiles_kind<Fused1024x8>(kind, dA, dB, tiles_a, tiles_b);
  if (variant == 4) return fused2_tiles_kind<Fused1024x8R32>(kind, dA, dB, tiles_a, tiles_b);
#endif
  if (dA.n == 960 || dA.n == 896) {
    if (kind != FUSED_COLS_ROWS) return -1;
    *tiles_a = (int)(dA.n == 960 ? Fused960::ColsToRing::ntiles(dA) : Fused896::ColsToRing::ntiles(dA));
    *tiles_b = (int)(dA.n == 960 ? Fused960::RowsFromRing::ntiles(dB) : Fused896::RowsFromRing::ntiles(dB));
    return 0;
  }
  if (dA.n != dB.n || (dA.n == 512 && variant == 5)) {
    *tiles_a = (int)(dA.n == 512 ? Fused512T32::ColsToRing::ntiles(dA) : Fused1024R32::ColsToRing::ntiles(dA));
    *tiles_b = (int)(dB.n == 512 ? Fused512T32::RowsFromRing::ntiles(dB) : Fused1024R32::RowsFromRing::ntiles(dB));
    return 0;
  }
  if (kind == FUSED_PLANES_2D_B || kind == FUSED_PLANES_CR_B) {
    // (same tile shapes as kinds 3 / 1: the block jump changes addresses, not tiles)
    const int k = kind == FUSED_PLANES_2D_B ? FUSED_PLANES_2D : FUSED_COLS_ROWS;
    return dA.n == 512 ? fused2_tiles_kind<Fused512R32>(k, dA, dB, tiles_a, tiles_b) : fused2_tiles_kind<Fused1024R32>(k, dA, dB, tiles_a, tiles_b);
  }
  if (dA.n == 512) return fused2_tiles_kind<Fused512R32>(kind, dA, dB, tiles_a, tiles_b);
  return fused2_tiles_kind<Fused1024R32>(kind, dA, dB, tiles_a, tiles_b);
}

hipError_t launch_fused2_f64(int kind, int variant, const PassDesc &dA, const PassDesc &dB, const PassDesc *dev_descs,
                             const FusedDesc &f, const void *in, void *ring, void *out, hipStream_t s) {
#ifdef GFFT_VARIANTS
  if (variant == 2) return launch_fused2_kind<Fused1024x8>(kind, dA, dB, dev_descs, f, in, ring, out, s);
  if (variant == 4) return launch_fused2_kind<Fused1024x8R32>(kind, dA, dB, dev_descs, f, in, ring, out, s);
#endif
  if (dA.n == 960 && kind == FUSED_COLS_ROWS) return launch_fused2<Fused960::ColsToRing, Fused960::RowsFromRing>(dA, dB, dev_descs, f, in, ring, out, s);
  if (dA.n == 896 && kind == FUSED_COLS_ROWS) return launch_fused2<Fused896::ColsToRing, Fused896::RowsFromRing>(dA, dB, dev_descs, f, in, ring, out, s);
  if (dA.n == 960 || dA.n == 896) return hipErrorInvalidValue;
  if (dA.n == 512 && dB.n == 512 && variant == 5)
    return kind == FUSED_PLANES_CR_B ? launch_fused2<Fused512T32::ColsToRingB, Fused512T32::RowsFromRing>(dA, dB, dev_descs, f, in, ring, out, s)
                                     : launch_fused2<Fused512T32::ColsToRing, Fused512T32::RowsFromRing>(dA, dB, dev_descs, f, in, ring, out, s);
  if (dA.n != dB.n) {
    if (kind != FUSED_COLS_ROWS && kind != FUSED_PLANES_CR_B) return hipErrorInvalidValue;
    const bool blk = kind == FUSED_PLANES_CR_B;
    if (dA.n == 512)
      return blk ? launch_fused2<Fused512T32::ColsToRingB, Fused1024R32::RowsFromRing>(dA, dB, dev_descs, f, in, ring, out, s)
                 : launch_fused2<Fused512T32::ColsToRing, Fused1024R32::RowsFromRing>(dA, dB, dev_descs, f, in, ring, out, s);
    return blk ? launch_fused2<Fused1024R32::ColsToRingB, Fused512T32::RowsFromRing>(dA, dB, dev_descs, f, in, ring, out, s)
               : launch_fused2<Fused1024R32::ColsToRing, Fused512T32::RowsFromRing>(dA, dB, dev_descs, f, in, ring, out, s);
  }
  if (kind == FUSED_PLANES_2D_B)
    return dA.n == 512 ? launch_fused2<Fused512R32::RowsToRing, Fused512R32::ColsFromRingB>(dA, dB, dev_descs, f, in, ring, out, s)
                       : launch_fused2<Fused1024R32::RowsToRing, Fused1024R32::ColsFromRingB>(dA, dB, dev_descs, f, in, ring, out, s);
  if (kind == FUSED_PLANES_CR_B)
    return dA.n == 512 ? launch_fused2<Fused512R32::ColsToRingB, Fused512R32::RowsFromRing>(dA, dB, dev_descs, f, in, ring, out, s)
                       : launch_fused2<Fused1024R32::ColsToRingB, Fused1024R32::RowsFromRing>(dA, dB, dev_descs, f, in, ring, out, s);
  if (dA.n == 512) return launch_fused2_kind<Fused512R32>(kind, dA, dB, dev_descs, f, in, ring, out, s);
  return launch_fused2_kind<Fused1024R32>(kind, dA, dB, dev_descs, f, in, ring, out, s);
}

}  // namespace gfft
