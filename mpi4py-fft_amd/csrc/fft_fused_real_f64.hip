// fp64 fused pass pairs of REAL 3-D transforms (fft_pow2_impl.h fft_fused2_kernel; the complex pairs: fft_fused_f64.hip):
//   FUSED_R2C_PLANES  [packed-real r2c rows of 2 N reals -> strided n]   passes 1 + 2 of the r2c schedule, plane = one i0:
//                     its n1 rows are contiguous in the caller's array, the half spectrum (N + 1 entries, the row's last
//                     line completed with zeros) goes into the slot, the strided pass takes it along axis 1
//   FUSED_COLS_C2R    [strided n -> packed-real c2r rows]                passes 2 + 3 of the c2r schedule, plane = one i1
// The reference's default dtype is `float` (mpifft.py:202), i.e. these are the transforms a PFFT runs unless told otherwise
// (libfft.py:48-79, fftw/xfftn.py:173-326: the Hermitian axis is the last one, N / 2 + 1 entries).
// Rows: 16 values per thread, a row inside one wave (exchanges without barriers); strided: 32 values per thread, one
// exchange; 512 threads both, one workgroup per CU.
#include "fft_fused_impl.h"
#include <cstdlib>

namespace gfft {

// option c2r_2048: the c2r pair on rows of 2048 reals.  Round 4 measured it losing -- (1024,1024,2048) backward 21.4 -> 28.1 ms,
// every memory phase of its tiles 2-3 x slower than at 1024 reals -- and left it off, unexplained.  Round 5: the workspace
// pitch of that shape was 1032 = 8 x 129 entries, the one multiplier the channel hash folds onto itself (plan.cpp
// plan_fused3, pitch129); on 1040 entries the pair runs 11.8 ms against 5.8 + 6.7 for its two passes, the step 37.85 ->
// 36.33 ms (profiles/r05_ab_pitch129.txt).
int g_c2r_2048 = 1;

//                            real    N    R   T  COLS   SPLIT FLAGS                 MODE        BIGTW  radices
typedef PassCfg<double, 512, 16, 16, false, true, 1 | 2048 | 8192, MODE_R2C_H, false, 16, 8, 4> R2CRows512ToRing;      // 1024 reals per row
typedef PassCfg<double, 1024, 16, 8, false, true, 1 | 2048 | 8192, MODE_R2C_H, false, 16, 16, 4> R2CRows1024ToRing;    // 2048 reals per row
typedef PassCfg<double, 512, 16, 16, false, true, 2 | 4096 | 8192, MODE_C2R_H, false, 16, 8, 4> C2RRows512FromRing;
typedef PassCfg<double, 1024, 16, 8, false, true, 2 | 4096 | 8192, MODE_C2R_H, false, 16, 16, 4> C2RRows1024FromRing;     // 2048 reals per row (option c2r_2048)
typedef PassCfg<double, 1024, 32, 16, true, true, 1 | 8 | 2048 | 8192, MODE_C2C, false, 32, 32> Cols1024ToRing;
typedef PassCfg<double, 1024, 32, 16, true, true, 2 | 8 | 4096 | 8192, MODE_C2C, false, 32, 32> Cols1024FromRing;

bool fused2_real_supported_f64(int kind, int n_a, int n_b) {
  if (kind == FUSED_R2C_PLANES) return (n_a == 512 || n_a == 1024) && n_b == 1024;
  if (kind == FUSED_COLS_C2R && n_a == 1024 && n_b == 1024) return g_c2r_2048 != 0;
  if (kind == FUSED_COLS_C2R) return n_a == 1024 && n_b == 512;
  return false;
}

int fused2_real_tiles_f64(int kind, const PassDesc &dA, const PassDesc &dB, int *ta, int *tb) {
  if (kind == FUSED_R2C_PLANES) {
    *ta = (int)(dA.n == 512 ? R2CRows512ToRing::ntiles(dA) : R2CRows1024ToRing::ntiles(dA));
    *tb = (int)Cols1024FromRing::ntiles(dB);
    return 0;
  }
  if (kind == FUSED_COLS_C2R) {
    *ta = (int)Cols1024ToRing::ntiles(dA);
    if (dB.n == 1024) { *tb = (int)C2RRows1024FromRing::ntiles(dB); return 0; }
    *tb = (int)C2RRows512FromRing::ntiles(dB);
    return 0;
  }
  return -1;
}

hipError_t launch_fused2_real_f64(int kind, const PassDesc &dA, const PassDesc &dB, const PassDesc *dev, const FusedDesc &f,
                                  const void *in, void *ring, void *out, hipStream_t s) {
  if (kind == FUSED_R2C_PLANES) {
    if (dA.n == 512) return launch_fused2<R2CRows512ToRing, Cols1024FromRing>(dA, dB, dev, f, in, ring, out, s);
    return launch_fused2<R2CRows1024ToRing, Cols1024FromRing>(dA, dB, dev, f, in, ring, out, s);
  }
  if (kind == FUSED_COLS_C2R) {
    if (dB.n == 1024) return launch_fused2<Cols1024ToRing, C2RRows1024FromRing>(dA, dB, dev, f, in, ring, out, s);
    return launch_fused2<Cols1024ToRing, C2RRows512FromRing>(dA, dB, dev, f, in, ring, out, s);
  }
  return hipErrorInvalidValue;
}

}  // namespace gfft
