// fp64 fused pass pairs: two dependent passes of a transform in one persistent launch, the intermediate
// handed over through a ring buffer that lives in the Infinity Cache (fft_pow2_impl.h, fft_fused2_kernel).
//   [rows n] -> [strided n]   passes 1 + 2 of a complex 3-D schedule, forward order (plan.cpp plan_fused3; built, not the default)
//   [strided n] -> [rows n]   passes 2 + 3 of the complex 3-D schedule as both directions run it
//   four-step n x n           both passes of a length-n^2 transform (n = 1024: BASELINE config C2), second pass strided or rows
// Every pass is the plan of the stand-alone tables (fft_pow2_f64.hip) on 1024-thread workgroups -- the row
// pass therefore takes 16 rows per workgroup instead of 4 -- with the hand-off side at system scope.
#include "fft_fused_impl.h"

namespace gfft {

//                                  real   N     R   T   COLS   SPLIT FLAGS                 MODE      BIGTW  radices
// (FLAGS 1 / 2: the streams that are NOT the hand-off -- A's loads, B's stores -- are non-temporal, so that
// they do not push the ring out of the Infinity Cache: 1024^3 per step 37.35 -> 36.26 ms, clean A/B)
template <> struct FusedCfgs<double, 1024> {
  typedef PassCfg<double, 1024, 16, 16, false, true, 1 | 2048 | 8192, MODE_C2C, false, 16, 16, 4> RowsToRing;
  typedef PassCfg<double, 1024, 16, 16, false, true, 2 | 4096 | 8192, MODE_C2C, false, 16, 16, 4> RowsFromRing;
  typedef PassCfg<double, 1024, 16, 16, true, true, 1 | 8 | 2048 | 8192, MODE_C2C, false, 16, 16, 4> ColsToRing;
  typedef PassCfg<double, 1024, 16, 16, true, true, 2 | 8 | 4096 | 8192, MODE_C2C, false, 16, 16, 4> ColsFromRing;
  typedef PassCfg<double, 1024, 16, 16, true, true, 1 | 32 | 2048 | 8192, MODE_C2C, true, 16, 16, 4> FourStepFirst;
  // the four-step pair the other way round: the first pass stores its columns where they are (twiddled), the
  // second one reads the intermediate as ROWS -- exchanges inside the wave, no barriers -- and transposes on store
  typedef PassCfg<double, 1024, 16, 16, true, true, 1 | 2048 | 8192, MODE_C2C, true, 16, 16, 4> FourStepFirstNat;
  typedef PassCfg<double, 1024, 16, 16, false, true, 2 | 32 | 4096 | 8192, MODE_C2C, false, 16, 16, 4> RowsFromRingT;
};
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
};
#endif

bool fused2_supported_f64(int kind, int variant, int n_a, int n_b) {
  (void)kind;
  if (n_a != n_b) return false;
#ifdef GFFT_VARIANTS
  if (variant == 2) return n_a == 1024 && kind != FUSED_FOURSTEP_ROWS;
#endif
  return variant == 1 && n_a == 1024;
}

int fused2_tiles_f64(int kind, int variant, const PassDesc &dA, const PassDesc &dB, int *tiles_a, int *tiles_b) {
#ifdef GFFT_VARIANTS
  if (variant == 2) return fused2_tiles_kind<Fused1024x8>(kind, dA, dB, tiles_a, tiles_b);
#endif
  (void)variant;
  return fused2_tiles_kind<FusedCfgs<double, 1024>>(kind, dA, dB, tiles_a, tiles_b);
}

hipError_t launch_fused2_f64(int kind, int variant, const PassDesc &dA, const PassDesc &dB, const PassDesc *dev_descs,
                             const FusedDesc &f, const void *in, void *ring, void *out, hipStream_t s) {
#ifdef GFFT_VARIANTS
  if (variant == 2) return launch_fused2_kind<Fused1024x8>(kind, dA, dB, dev_descs, f, in, ring, out, s);
#endif
  (void)variant;
  return launch_fused2_kind<FusedCfgs<double, 1024>>(kind, dA, dB, dev_descs, f, in, ring, out, s);
}

}  // namespace gfft
