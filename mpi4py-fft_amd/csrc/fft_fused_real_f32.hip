// fp32 fused pass pairs of REAL 3-D transforms (the fp64 ones: fft_fused_real_f64.hip; the kernel: fft_pow2_impl.h
// fft_fused2_kernel): [packed-real r2c rows of 1024 reals -> strided n = 1024] on the contiguous planes i0 of the r2c
// schedule, [strided n = 1024 -> packed-real c2r rows] on the planes i1 of the c2r schedule -- 1024^3 real fp32, the
// single-GPU relative of BASELINE config C5.  1024-thread workgroups: rows 16 values per thread (c2r: 8, the R = 16
// c2r plan needs ~170 VGPRs), strided 32 values per thread on 32 columns (256-byte segments).
#include "fft_fused_impl.h"

namespace gfft {

//                           real   N    R   T  COLS   SPLIT  FLAGS                 MODE        BIGTW  radices
typedef PassCfg<float, 512, 16, 32, false, false, 1 | 2048 | 8192, MODE_R2C_H, false, 16, 8, 4> R2CRows512ToRingF32;
typedef PassCfg<float, 512, 8, 16, false, false, 2 | 4096 | 8192, MODE_C2R_H, false, 8, 8, 8> C2RRows512FromRingF32;
typedef PassCfg<float, 1024, 32, 32, true, true, 1 | 8 | 2048 | 8192, MODE_C2C, false, 16, 16, 4> Cols1024ToRingF32;
typedef PassCfg<float, 1024, 32, 32, true, true, 2 | 8 | 4096 | 8192, MODE_C2C, false, 16, 16, 4> Cols1024FromRingF32;

bool fused2_real_supported_f32(int kind, int n_a, int n_b) {
  if (kind == FUSED_R2C_PLANES) return n_a == 512 && n_b == 1024;
  if (kind == FUSED_COLS_C2R) return n_a == 1024 && n_b == 512;
  return false;
}

int fused2_real_tiles_f32(int kind, const PassDesc &dA, const PassDesc &dB, int *ta, int *tb) {
  if (kind == FUSED_R2C_PLANES) { *ta = (int)R2CRows512ToRingF32::ntiles(dA); *tb = (int)Cols1024FromRingF32::ntiles(dB); return 0; }
  if (kind == FUSED_COLS_C2R) { *ta = (int)Cols1024ToRingF32::ntiles(dA); *tb = (int)C2RRows512FromRingF32::ntiles(dB); return 0; }
  return -1;
}

hipError_t launch_fused2_real_f32(int kind, const PassDesc &dA, const PassDesc &dB, const PassDesc *dev, const FusedDesc &f,
                                  const void *in, void *ring, void *out, hipStream_t s) {
  if (kind == FUSED_R2C_PLANES) return launch_fused2<R2CRows512ToRingF32, Cols1024FromRingF32>(dA, dB, dev, f, in, ring, out, s);
  if (kind == FUSED_COLS_C2R) return launch_fused2<Cols1024ToRingF32, C2RRows512FromRingF32>(dA, dB, dev, f, in, ring, out, s);
  return hipErrorInvalidValue;
}

}  // namespace gfft
