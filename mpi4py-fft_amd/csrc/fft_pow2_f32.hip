// fp32 (complex64) instantiations of the power-of-two pass kernels (the `fftwf_*` clone of the
// reference, setup.py:93-111).  A complex64 is 8 bytes, so R = 16 costs the registers R = 8
// costs in fp64, and 16 adjacent columns make the 128-byte segment.
#include "fft_pow2_impl.h"
#ifdef GFFT_VARIANTS
#define GFFT_HAS_VARIANTS 1
#else
#define GFFT_HAS_VARIANTS 0
#endif

namespace gfft {

#define P32(N, R, T, COLS, SPLIT, MINW, ...) \
  launch_pow2_inst<float, N, R, T, COLS, SPLIT, MINW, 0, __VA_ARGS__>(d, in, out, s)
#define P32F(N, R, T, COLS, SPLIT, MINW, FLAGS, ...) \
  launch_pow2_inst<float, N, R, T, COLS, SPLIT, MINW, FLAGS, __VA_ARGS__>(d, in, out, s)

bool pow2_supported_f32(int n) { return n >= 16 && n <= 4096 && (n & (n - 1)) == 0; }

hipError_t launch_pow2_f32(const PassDesc &d, bool cols, int variant, const void *in, void *out,
                           hipStream_t s) {
  if (!cols) {
    // (the non-temporal kernels below: plain complex passes over arrays that do not fit the Infinity Cache -- on 128 MiB of traffic they LOSE 10-30 %)
    const bool plain = variant == 0 && d.mode == MODE_C2C && !d.tw_hi && !d.tr_dir && 2.0 * (double)d.batch * d.n * sizeof(float) * 2 >= 268435456.0;
    switch (d.n) {
      case 16: return P32(16, 4, 16, false, false, 1, 4, 4);
      case 32: return P32(32, 8, 16, false, false, 1, 8, 4);
      case 64: return P32(64, 8, 8, false, false, 1, 8, 8);
      case 128: return P32(128, 8, 4, false, false, 1, 8, 8, 2);
      case 256: return P32(256, 16, 4, false, false, 1, 16, 16);
      // Round 6 (as fft_pow2_f64.hip): plain complex row passes with non-temporal loads and stores; n = 512 on 16 values per thread, 8 rows per 256
      // threads (512^3 0.401 -> 0.359 ms, (2048,512,512) 1.667 -> 1.535 ms; ONE exchange -- 32 values per thread -- level on the large array, +8 % on
      // 512^3), n = 1024 on 4 rows per workgroup instead of 1 ((512,1024,1024) 1.678 -> 1.518 ms); n = 2048 level (profiles/r06_rows_probe.txt).
      // variant 16 = the former table.
      case 512:
        if (plain) return P32F(512, 16, 8, false, false, 1, 8 | 3, 16, 8, 4);
        return P32(512, 8, 1, false, false, 1, 8, 8, 8);
      case 1024:
        if (plain) return P32F(1024, 16, 4, false, false, 1, 8 | 3, 16, 16, 4);
        return P32(1024, 16, 1, false, false, 1, 16, 16, 4);
      case 2048: return P32(2048, 16, 1, false, false, 1, 16, 16, 8);
      case 4096: return P32(4096, 16, 1, false, false, 1, 16, 16, 16);
    }
  } else if (d.tw_hi && d.out_es == 1 && d.mode == MODE_C2C && !d.tr_dir && d.n >= 512 && variant != 9) {
    // first pass of a four-step transform: four-step twiddle, then a transposing store through LDS so
    // that each output line is written in whole rows (as in fft_pow2_f64.hip; measured on 128 x 2^20
    // c64 under rocprofv3: this pass took 934 us with the plain store, the second pass 501 us)
    switch (d.n) {
      case 512: return P32F(512, 16, 32, true, true, 1, 32, 16, 8, 4);
      case 1024: return P32F(1024, 16, 16, true, true, 1, 32, 16, 16, 4);
      case 2048: return P32F(2048, 16, 8, true, true, 4, 32, 16, 16, 8);
      case 4096: return P32F(4096, 16, 4, true, true, 4, 32, 16, 16, 16);
    }
  } else if (d.mode == MODE_C2C && d.tr_dir && !d.tw_hi && (d.n == 512 || (GFFT_HAS_VARIANTS && variant == 7 && d.n >= 1024))) {
    // complex strided passes with fused truncation (store side) / zero padding (load side) on the
    // 256-byte tiles of the plain passes below.  n = 512 compiles clean (100 / 82 VGPRs).  From
    // n = 1024 the R = 32 plans fit 128 VGPRs only with 9-16 registers of scratch (145 before the
    // line was pinned between the last stage and the stores) and measured SLOWER than the shared
    // table's 128-byte tiles -- (1024,1024,2048) c64, padded axis 1: 7.9 / 10.2 ms forward /
    // backward against 7.2 / 7.0 ms -- so they stay behind variant 7 (A/B runs).
#define P32T(N, R, T, ...)                                                                                          \
  (d.tr_dir == 1 ? launch_pow2_one<float, N, R, T, true, true, 1, 16, MODE_C2C, false, __VA_ARGS__>(d, in, out, s) \
                 : launch_pow2_one<float, N, R, T, true, true, 1, 16 | 64, MODE_C2C, false, __VA_ARGS__>(d, in, out, s))
    switch (d.n) {
      case 512: return P32T(512, 16, 32, 16, 8, 4);
#ifdef GFFT_VARIANTS   // (variant 7 only: measured slower, see above)
      case 1024: return P32T(1024, 32, 32, 16, 16, 4);
      case 2048: return P32T(2048, 32, 16, 16, 16, 8);
      case 4096: return P32T(4096, 32, 8, 16, 16, 16);
#endif
    }
#undef P32T
  } else if (d.mode != MODE_C2C || d.tw_hi || d.tr_dir || d.out_es == 1 || d.in_es == 1) {
    // real modes along a strided axis, fused truncation / padding, four-step passes: the widest
    // tile whose every mode compiles without spills (R = 16 up to n = 4096; the four-step twiddle
    // variant only up to n = 1024); variant 9 = the former lean R = 8 plans
    if (variant != 9 && !(d.tw_hi && d.n > 1024)) {
      switch (d.n) {
        case 64: return P32(64, 8, 32, true, false, 1, 8, 8);
        case 128: return P32(128, 8, 32, true, false, 1, 8, 8, 2);
        case 256: return P32(256, 8, 32, true, false, 1, 8, 8, 4);
        case 512: return P32(512, 16, 32, true, true, 1, 16, 8, 4);
        case 1024: return P32(1024, 16, 16, true, true, 1, 16, 16, 4);
        case 2048: return P32(2048, 16, 8, true, true, 4, 16, 16, 8);
        case 4096: return P32(4096, 16, 4, true, true, 4, 16, 16, 16);
      }
    }
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
    // Round 6: non-temporal loads and stores at n = 1024 / 2048 where every row starts on a 128-byte line and the array is beyond the Infinity
    // Cache -- natural-stride stage arrays (the one-rank schedules ask for variant 2): (1024,256,1024) axis 0 1.183 -> 1.056 ms, (256,1024,1024)
    // axis 1 0.990 -> 0.953 ms, (2048,512,512) axis 0 -- the slower twin of config C5's last stage -- 2.997 -> 2.780 ms, (512,2048,512) axis 1
    // 2.098 -> 2.027 ms.  On 513-wide rows the same streams cost up to 65 % ((512,2048,513) axis 1 2.72 -> 4.48 ms: partial lines written
    // around the cache): they keep the plain ones (profiles/r06_cols_nt_probe.txt).
    const bool whole = strided_lines_whole(d, 8);
    const bool nt_ok = whole && d.inner % 16 == 0 && 2.0 * (double)d.batch * d.n * 8 >= 268435456.0;
    // plain c2c along a strided axis: 32 adjacent columns = 256-byte segments up to n = 512
    // (measured on (2048,512,1024) c64: 3.88 -> 3.35 ms; n = 256: 3.53 -> 3.37 ms).
    switch (d.n) {
      case 16: return P32F(16, 4, 32, true, false, 1, 8, 4, 4);
      case 32: return P32F(32, 8, 32, true, false, 1, 8, 8, 4);
      case 64: return P32F(64, 8, 32, true, false, 1, 8, 8, 8);
      case 128: return P32F(128, 8, 32, true, false, 1, 8, 8, 8, 2);
      case 256:
        // Round 6: 32 values per thread = ONE exchange (radices 32 x 8), the same 32 columns = 256-byte segments, 256 threads.  Half the bytes
        // per butterfly of complex128: the exchange phase is what complex64 tiles wait on, and this halves it.  Plans
        // alternating on the same arrays (tools/cols_variant_probe.py, profiles/r06_cols_t32_probe.txt): (256,256,256) axis 1 / 0
        // 0.068 / 0.065 -> 0.058 / 0.058 ms, (1024,256,1024) axis 1 0.893 -> 0.796 ms, (256,1024,1024) axis 0 1.126 -> 0.836 ms.
        // (64 columns on 512 or 1024 threads: measured behind except on the smallest array.)  Odd widths keep the three-stage tile: see n = 512.
        // Inside one-rank 3-D schedules (plan_fused3 asks for variant 2 there) too: 256^3 c64 per step 0.363 -> 0.320 ms; the n = 512 lines of
        // such schedules keep variant 2 (512^3 c64: 2.442 ms against 2.547 with this tile shape).
        if ((variant == 0 || variant == 2) && d.inner % 32 == 0 && whole && 2.0 * (double)d.batch * 256 * 8 >= 268435456.0) return P32F(256, 32, 32, true, true, 2, 8 | 3, 32, 8);      // (non-temporal: arrays beyond the Infinity Cache)
        switch (variant) {
          default: return P32F(256, 8, 32, true, false, 1, 8, 8, 8, 4);
          case 22: return P32F(256, 32, 32, true, true, 2, 8 | 3, 32, 8);     // (the automatic choice above, whatever the array: for A/B)
#ifdef GFFT_VARIANTS
          case 1: return P32(256, 16, 16, true, false, 1, 16, 16);
#endif
        }
      case 512:
        // Round 6: 32 values per thread = ONE exchange (radices 32 x 16), 32 columns, 512 threads: (512,512,512) axis 1 / 0 0.500 / 0.588 ->
        // 0.435 / 0.449 ms, (1024,512,1024) axis 1 1.862 -> 1.795 ms, (512,1024,1024) axis 0 2.409 -> 2.080 ms -- where rows are whole
        // multiples of the tile.  On 513-wide rows (the half spectra of real transforms) it LOSES: (2048,512,513) axis 1 2.353 -> 2.936 ms.
        // (64 columns = 512-byte segments on 1024 threads: measured behind the default everywhere.)
        if (variant == 0 && d.inner % 32 == 0 && whole && 2.0 * (double)d.batch * 512 * 8 >= 268435456.0) return P32F(512, 32, 32, true, true, 2, 8 | 3, 32, 16);
        switch (variant) {
          default: return P32F(512, 16, 32, true, true, 1, 8, 16, 8, 4);
          case 22: return P32F(512, 32, 32, true, true, 2, 8 | 3, 32, 16);    // (the automatic choice above, whatever the array: for A/B)
#ifdef GFFT_VARIANTS
          case 1: return P32(512, 8, 16, true, true, 1, 8, 8, 8);
#endif
          case 2: return P32F(512, 16, 16, true, true, 4, 8, 16, 8, 4);           // A/B as for n = 1024: 0.95 / 1.26 ms against 0.91 / 1.04
        }
      // n >= 1024: R = 32 elements per thread (the 64 data VGPRs R = 16 costs in fp64) doubles the
      // columns per workgroup at the same 1024 threads: 256-byte segments at n = 1024, 128 at 2048.
      // Measured ((.,n,1024) c64 axis 1, R = 16 -> R = 32): n=1024 4.03 -> 3.65 ms, n=2048
      // 5.83 -> 4.13 ms, n=4096 (x512) 4.85 -> 3.37 ms; variant 1 = the R = 16 plans.
      case 1024:
        switch (variant) {
          case 0: if (nt_ok) return P32F(1024, 32, 32, true, true, 1, 8 | 3, 16, 16, 4);      // (falls through to the plain streams otherwise)
          // (R6, measured and NOT kept: ONE exchange -- radices 32 x 32 -- on 16 columns = 128-byte segments, 512 threads of 182 VGPRs, two workgroups per CU,
          // non-temporal: near strides -2 ... -3.5 % ((256,1024,1024) axis 1 0.972 -> 0.938 ms), far strides +13 ... +15 % ((1024,256,1024) axis 0 1.058 -> 1.212 ms):
          // where the exchange halves, the segment width costs more)
          default: return P32F(1024, 32, 32, true, true, 1, 8, 16, 16, 4);
#ifdef GFFT_VARIANTS
          case 1: return P32(1024, 16, 16, true, true, 1, 16, 16, 4);
          case 8: return P32F(1024, 32, 32, true, true, 1, 8 | 256, 16, 16, 4);   // A/B: line not pinned before the stores
#endif
          // A/B: two 512-thread workgroups per CU (one computes while the other loads) on 128-byte
          // segments: 1024^3 c64 axis 1 3.60 ms (default 3.58), axis 0 5.38 (4.43) -- segment width wins.
          // Re-measured after the 4-byte LDS bank rule (tools/variant_probe_f32.py; this variant's T = 16
          // exchanges had been the conflicting ones): pitched rows near 3.40 (3.55) / far 3.90 (3.99),
          // natural rows near 3.50 (3.53) / far 4.94 (4.15) -- ahead by 2-4 % only where both sides are
          // pitched, behind by 19 % on natural far strides: still not the default.
          case 2: return P32F(1024, 32, 16, true, true, 4, 8, 16, 16, 4);
          // (R4: non-temporal loads and stores on either form, as the fp64 strided default has them: 1024^3 c64 per step 19.61 ->
          // 19.57 ms on variant 2, 20.43 on the wide tile; 1024^3 r2c f32 11.65 -> 12.12 / 12.87 ms -- not adopted)
#ifdef GFFT_VARIANTS
          // R4, measured and NOT kept: 64 values per thread, radices 64 x 16 = ONE exchange, 512 threads on 32 columns
          // (256-byte segments) -- what paid in fp64 (32 values, fft_pow2_f64.hip) does not here: 64 complex64 are 128 VGPRs of
          // data alone, the kernels take all 256 plus 130-160 bytes of scratch, and (1024,1024,1024) axis 1 goes 3.82 -> 4.72 ms
          // (4.50 with non-temporal streams), axis 0 4.8-4.9 -> 6.44 / 4.93 ms (profiles/r04_variant_cols_f32_r64.txt)
          case 5: return P32F(1024, 64, 32, true, true, 2, 8, 64, 16);
          case 6: return P32F(1024, 64, 32, true, true, 2, 8 | 3, 64, 16);     // ... with non-temporal loads and stores
          case 10: return P32F(1024, 32, 32, true, true, 1, 8 | 4, 16, 16, 4);    // R6: the ACCESS PATTERN ALONE of the default tile (tools/strided_bound_probe_f32.py)
          // (R6, measured and NOT kept: touches of the NEXT tile's lines issued behind this tile's loads -- one dword per element into a VGPR nobody reads --
          // so that HBM works during the exchange phase: near strides +8 ... +10 % SLOWER, far strides +-1 %: the touches cross the L2 boundary too,
          // which is what the launch is short of; profiles/r06_touch_probe.txt)
#endif
#ifdef GFFT_VARIANTS
          // A/B: two radix-32 stages = ONE exchange instead of two (LDS cycles and barriers halved), but the
          // 32-point butterfly with its 31 stage twiddles does not fit 128 VGPRs at 1024 threads
          // (116 B of scratch per lane): pitched near 4.33 ms against 3.61, far 5.18 against 4.02
          case 3: return P32F(1024, 32, 32, true, true, 1, 8, 32, 32);
#endif
        }
      case 2048:
        switch (variant) {
          case 0: if (nt_ok) return P32F(2048, 32, 16, true, true, 1, 8 | 3, 16, 16, 8);      // (falls through to the plain streams otherwise)
          default: return P32F(2048, 32, 16, true, true, 1, 8, 16, 16, 8);
#ifdef GFFT_VARIANTS
          // R4, measured and NOT kept (as at n = 1024): radices 64 x 32 = ONE exchange, 512 threads on 16 columns -- the C5 stages
          // (512,2048,513) axis 1 2.53 -> 2.80 ms (3.81 with non-temporal streams), (2048,512,513) axis 0 2.38 -> 2.65 / 2.36 ms
          case 5: return P32F(2048, 64, 16, true, true, 2, 8, 64, 32);
          case 6: return P32F(2048, 64, 16, true, true, 2, 8 | 3, 64, 32);
          case 10: return P32F(2048, 32, 16, true, true, 1, 8 | 4, 16, 16, 8);    // R6: access pattern alone
#endif
#ifdef GFFT_VARIANTS
          case 1: return P32(2048, 16, 8, true, true, 4, 16, 16, 8);
#endif
        }
      case 4096:
        switch (variant) {
          default: return P32F(4096, 32, 8, true, true, 1, 8, 16, 16, 16);
#ifdef GFFT_VARIANTS
          case 1: return P32(4096, 16, 4, true, true, 4, 16, 16, 16);
#endif
        }
    }
  }
  return hipErrorInvalidValue;
}

hipError_t launch_pow2_r2r_f32(const PassDesc &d, bool cols, const void *in, void *out, hipStream_t s) {
  if (!cols) {
    switch (d.n) {
      case 64: return launch_pow2_one<float, 64, 8, 8, false, false, 1, 0, MODE_R2R, false, 8, 8>(d, in, out, s);
      case 128: return launch_pow2_one<float, 128, 8, 4, false, false, 1, 0, MODE_R2R, false, 8, 8, 2>(d, in, out, s);
      case 256: return launch_pow2_one<float, 256, 16, 4, false, false, 1, 0, MODE_R2R, false, 16, 16>(d, in, out, s);
      case 512: return launch_pow2_one<float, 512, 8, 1, false, false, 1, 0, MODE_R2R, false, 8, 8, 8>(d, in, out, s);
      case 1024: return launch_pow2_one<float, 1024, 16, 1, false, false, 1, 0, MODE_R2R, false, 16, 16, 4>(d, in, out, s);
      case 2048: return launch_pow2_one<float, 2048, 16, 1, false, false, 1, 0, MODE_R2R, false, 16, 16, 8>(d, in, out, s);
      case 4096: return launch_pow2_one<float, 4096, 16, 1, false, false, 1, 0, MODE_R2R, false, 16, 16, 16>(d, in, out, s);
    }
  } else {
    switch (d.n) {
      case 64: return launch_pow2_one<float, 64, 8, 32, true, false, 1, 0, MODE_R2R, false, 8, 8>(d, in, out, s);
      case 128: return launch_pow2_one<float, 128, 8, 32, true, false, 1, 0, MODE_R2R, false, 8, 8, 2>(d, in, out, s);
      case 256: return launch_pow2_one<float, 256, 8, 32, true, false, 1, 0, MODE_R2R, false, 8, 8, 4>(d, in, out, s);
      case 512: return launch_pow2_one<float, 512, 16, 32, true, true, 1, 0, MODE_R2R, false, 16, 8, 4>(d, in, out, s);
      case 1024: return launch_pow2_one<float, 1024, 16, 16, true, true, 1, 0, MODE_R2R, false, 16, 16, 4>(d, in, out, s);
      case 2048: return launch_pow2_one<float, 2048, 16, 8, true, true, 1, 0, MODE_R2R, false, 16, 16, 8>(d, in, out, s);
      case 4096: return launch_pow2_one<float, 4096, 16, 4, true, true, 1, 0, MODE_R2R, false, 16, 16, 16>(d, in, out, s);
    }
  }
  return hipErrorInvalidValue;
}

}  // namespace gfft
