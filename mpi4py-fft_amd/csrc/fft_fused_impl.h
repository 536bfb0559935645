// Fused pass pairs: the launch tables shared by fft_fused_f64.hip / fft_fused_f32.hip.  A pair exists for a
// (precision, n) when both passes -- the row plan and the strided plan of the stand-alone tables
// (fft_pow2_f64.hip / fft_pow2_f32.hip), rebuilt on 1024-thread workgroups -- fit one workgroup shape.
#pragma once
#include "fft_pow2_impl.h"

namespace gfft {

// FLAGS of the hand-off sides: 2048 = stores at system scope, 4096 = loads at system scope, 8192 = natural
// layouts (descriptor layout fields are compile-time zeros); 8 = plain complex strided pass, 32 = first
// four-step pass (twiddle + transposing store)
template <typename real, int N> struct FusedCfgs;

template <typename C>
static hipError_t launch_fused2_kind(int kind, const PassDesc &dA, const PassDesc &dB, const PassDesc *dev, const FusedDesc &f,
                                     const void *in, void *ring, void *out, hipStream_t s) {
  switch (kind) {
    case FUSED_PLANES_2D:
    case FUSED_ROWS_COLS: return launch_fused2<typename C::RowsToRing, typename C::ColsFromRing>(dA, dB, dev, f, in, ring, out, s);
    case FUSED_COLS_ROWS: return launch_fused2<typename C::ColsToRing, typename C::RowsFromRing>(dA, dB, dev, f, in, ring, out, s);
    case FUSED_FOURSTEP: return launch_fused2<typename C::FourStepFirst, typename C::ColsFromRing>(dA, dB, dev, f, in, ring, out, s);
    case FUSED_FOURSTEP_ROWS: return launch_fused2<typename C::FourStepFirstNat, typename C::RowsFromRingT>(dA, dB, dev, f, in, ring, out, s);
  }
  return hipErrorInvalidValue;
}

// tiles per plane of either pass: rows and four-step first passes tile the flat batch, strided passes the
// columns of each row of the batch
template <typename C>
static int fused2_tiles_kind(int kind, const PassDesc &dA, const PassDesc &dB, int *tiles_a, int *tiles_b) {
  switch (kind) {
    case FUSED_PLANES_2D:
    case FUSED_ROWS_COLS: *tiles_a = (int)C::RowsToRing::ntiles(dA); *tiles_b = (int)C::ColsFromRing::ntiles(dB); return 0;
    case FUSED_COLS_ROWS: *tiles_a = (int)C::ColsToRing::ntiles(dA); *tiles_b = (int)C::RowsFromRing::ntiles(dB); return 0;
    case FUSED_FOURSTEP: *tiles_a = (int)C::FourStepFirst::ntiles(dA); *tiles_b = (int)C::ColsFromRing::ntiles(dB); return 0;
    case FUSED_FOURSTEP_ROWS: *tiles_a = (int)C::FourStepFirstNat::ntiles(dA); *tiles_b = (int)C::RowsFromRingT::ntiles(dB); return 0;
  }
  return -1;
}

}  // namespace gfft
