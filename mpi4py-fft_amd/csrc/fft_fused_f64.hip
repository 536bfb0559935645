// fp64 fused pass pairs: two dependent passes of a transform in one persistent launch, the intermediate
// handed over through a ring buffer that lives in the Infinity Cache (fft_pow2_impl.h, fft_fused2_kernel).
//   [rows n = 1024] -> [strided n = 1024]   passes 1 + 2 of the single-GPU 3-D schedule, forward (plan.cpp plan_fused3)
//   [strided n = 1024] -> [rows n = 1024]   passes 2 + 3 of the backward schedule
//   four-step 1024 x 1024                    both passes of a length-2^20 transform (BASELINE config C2)
// Every pass is the plan of the stand-alone tables (fft_pow2_f64.hip) on 1024-thread workgroups -- the row
// pass therefore takes 16 rows per workgroup instead of 4 -- with the hand-off side at system scope.
#include "fft_pow2_impl.h"

namespace gfft {

namespace {
//                      real   N     R   T   COLS   SPLIT FLAGS        MODE      BIGTW  radices
typedef PassCfg<double, 1024, 16, 16, false, true, 2048 | 8192, MODE_C2C, false, 16, 16, 4> RowsToRing;
typedef PassCfg<double, 1024, 16, 16, false, true, 4096 | 8192, MODE_C2C, false, 16, 16, 4> RowsFromRing;
typedef PassCfg<double, 1024, 16, 16, true, true, 8 | 2048 | 8192, MODE_C2C, false, 16, 16, 4> ColsToRing;
typedef PassCfg<double, 1024, 16, 16, true, true, 8 | 4096 | 8192, MODE_C2C, false, 16, 16, 4> ColsFromRing;
typedef PassCfg<double, 1024, 16, 16, true, true, 32 | 2048 | 8192, MODE_C2C, true, 16, 16, 4> FourStepFirst;   // twiddle + transposing store
// variant 2: 8 lines per tile, 512 threads, two workgroups per CU (a fused pair moves a third less through HBM
// than two launches, so what bounds it is how long ONE workgroup takes per tile -- load, butterflies and
// store in sequence -- and a second workgroup on the CU fills those gaps)
typedef PassCfg<double, 1024, 16, 8, false, true, 2048 | 8192, MODE_C2C, false, 16, 16, 4> RowsToRing8;
typedef PassCfg<double, 1024, 16, 8, false, true, 4096 | 8192, MODE_C2C, false, 16, 16, 4> RowsFromRing8;
typedef PassCfg<double, 1024, 16, 8, true, true, 8 | 2048 | 8192, MODE_C2C, false, 16, 16, 4> ColsToRing8;
typedef PassCfg<double, 1024, 16, 8, true, true, 8 | 4096 | 8192, MODE_C2C, false, 16, 16, 4> ColsFromRing8;
typedef PassCfg<double, 1024, 16, 8, true, true, 32 | 2048 | 8192, MODE_C2C, true, 16, 16, 4> FourStepFirst8;
}  // namespace

bool fused2_supported(int kind, int precision, int n_a, int n_b) {
  (void)kind;
  return precision == 8 && n_a == 1024 && n_b == 1024;
}

int fused2_tiles(int kind, int variant, const PassDesc &dA, const PassDesc &dB, int *tiles_a, int *tiles_b) {
  const int T = variant == 2 ? 8 : 16;
  (void)kind;
  // rows and four-step first passes tile the flat batch, strided passes the columns of each row of the batch
  auto flat = [&](const PassDesc &d) { return (int)((d.batch + T - 1) / T); };
  auto cols = [&](const PassDesc &d) { return (int)((d.batch / d.inner) * ((d.inner + T - 1) / T)); };
  switch (kind) {
    case FUSED_ROWS_COLS: *tiles_a = flat(dA); *tiles_b = cols(dB); return 0;
    case FUSED_COLS_ROWS: *tiles_a = cols(dA); *tiles_b = flat(dB); return 0;
    case FUSED_FOURSTEP: *tiles_a = flat(dA); *tiles_b = cols(dB); return 0;
  }
  return -1;
}

hipError_t launch_fused2_f64(int kind, int variant, const PassDesc &dA, const PassDesc &dB, const FusedDesc &f, const void *in,
                             void *ring, void *out, hipStream_t s) {
  if (variant == 2) {
    switch (kind) {
      case FUSED_ROWS_COLS: return launch_fused2<RowsToRing8, ColsFromRing8>(dA, dB, f, in, ring, out, s);
      case FUSED_COLS_ROWS: return launch_fused2<ColsToRing8, RowsFromRing8>(dA, dB, f, in, ring, out, s);
      case FUSED_FOURSTEP: return launch_fused2<FourStepFirst8, ColsFromRing8>(dA, dB, f, in, ring, out, s);
    }
    return hipErrorInvalidValue;
  }
  switch (kind) {
    case FUSED_ROWS_COLS: return launch_fused2<RowsToRing, ColsFromRing>(dA, dB, f, in, ring, out, s);
    case FUSED_COLS_ROWS: return launch_fused2<ColsToRing, RowsFromRing>(dA, dB, f, in, ring, out, s);
    case FUSED_FOURSTEP: return launch_fused2<FourStepFirst, ColsFromRing>(dA, dB, f, in, ring, out, s);
  }
  return hipErrorInvalidValue;
}

}  // namespace gfft
