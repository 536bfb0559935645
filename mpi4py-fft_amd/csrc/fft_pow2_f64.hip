// fp64 (complex128) instantiations of the power-of-two pass kernels.
//
// Plan table (R = elements per thread, T = columns per workgroup, radices per stage).
// Defaults at N = 1024: rows R = 16 / T = 4 (111 VGPRs), strided R = 16 / T = 16 = 256-byte
// segments (107 VGPRs, 1024 threads); the `variant` argument selects the measured alternatives
// (DESIGN_HISTORY.md section 4.1 has the A/B numbers that picked the defaults).
#include "fft_pow2_impl.h"

namespace gfft {

//                 real    N  R  T  COLS  SPLIT MINW radices
#define P64(N, R, T, COLS, MINW, ...) \
  launch_pow2_inst<double, N, R, T, COLS, true, MINW, 0, __VA_ARGS__>(d, in, out, s)
#define P64F(N, R, T, COLS, MINW, FLAGS, ...) \
  launch_pow2_inst<double, N, R, T, COLS, true, MINW, FLAGS, __VA_ARGS__>(d, in, out, s)

bool pow2_supported_f64(int n) { return n >= 16 && n <= 4096 && (n & (n - 1)) == 0; }

hipError_t launch_pow2_f64(const PassDesc &d, bool cols, int variant, const void *in, void *out,
                           hipStream_t s) {
  if (!cols) {
    // (the non-temporal kernels below: plain complex passes over arrays that do not fit the Infinity Cache -- on 128 MiB of traffic they LOSE 10 %)
    const bool plain = variant == 0 && d.mode == MODE_C2C && !d.tw_hi && !d.tr_dir && 2.0 * (double)d.batch * d.n * sizeof(double) * 2 >= 268435456.0;
    switch (d.n) {
      case 16: return P64(16, 4, 16, false, 1, 4, 4);
      case 32: return P64(32, 8, 16, false, 1, 8, 4);
      case 64: return P64(64, 8, 8, false, 1, 8, 8);
      case 128: return P64(128, 8, 4, false, 1, 8, 8, 2);
      // Round 6: row passes with NON-TEMPORAL loads and stores (the arrays are streamed once: nothing to keep in L2 / Infinity Cache), and at
      // n = 512 / 1024 on 32 values per thread = ONE exchange inside the wave (16 / 8 rows per 256 threads).  Plans alternating on the same
      // arrays (tools/cols_variant_probe.py with PROBE_OPT=variant_rows, profiles/r06_rows_probe.txt): n = 1024 (256,512,1024) -- config C4's
      // first stage on 8 GPUs -- 0.777 -> 0.722 ms, 1024^3 6.58 -> 6.33 ms; n = 512 512^3 0.789 -> 0.729 ms; n = 256 256^3 0.099 -> 0.088 ms,
      // n = 2048 (256,512,2048) 1.534 -> 1.420 ms (these two on the table's radices: one exchange measured behind there).  variant 16 = the
      // former table.
      case 256:
        if (plain) return P64F(256, 8, 8, false, 1, 8 | 3, 8, 8, 4);
        return P64(256, 8, 8, false, 1, 8, 8, 4);
      case 512:
        if (plain) return P64F(512, 32, 16, false, 2, 8 | 3, 32, 16);
        return P64(512, 8, 4, false, 1, 8, 8, 8);
      case 1024:
        // fused zero-padding on load / truncation on store: the lean plan (measured on the padded
        // 683^3 -> 1024^3 backward row pass: 8.4 ms with R = 16, 6.7 ms with R = 8)
        if (d.tr_dir && variant == 0) return P64(1024, 8, 2, false, 1, 8, 8, 8, 2);
        if (plain) return P64F(1024, 32, 8, false, 2, 8 | 3, 32, 32);        // one exchange inside the wave, 8 rows / 256 threads, non-temporal
        switch (variant) {
          default: return P64(1024, 16, 4, false, 1, 16, 16, 4);   // 4 rows / 256 threads, 2 exchanges (the table through round 5; fused truncation / padding, r2c / c2r)
          // (R4: 32 values per thread / ONE exchange inside the wave, 8 rows per 256 threads, was measured too: level --
          // (256,512,1024) axis 2 0.745-0.776 against 0.760-0.764 ms; row passes already run at the copy rate)
#ifdef GFFT_VARIANTS   // measured alternatives (make VARIANTS=1): not in the shipped library
          case 1: return P64(1024, 8, 1, false, 1, 8, 8, 8, 2);
          case 2: return P64(1024, 16, 1, false, 1, 16, 16, 4);
          case 4: return P64F(1024, 8, 2, false, 1, 3, 8, 8, 8, 2);      // nt loads+stores
          case 5: return P64F(1024, 8, 2, false, 1, 4, 8, 8, 8, 2);      // access pattern only
          case 6: return P64F(1024, 8, 2, false, 1, 7, 8, 8, 8, 2);      // access pattern, nt
          case 7: return P64F(1024, 8, 2, false, 1, 1, 8, 8, 8, 2);      // nt loads
          case 8: return P64F(1024, 8, 2, false, 1, 2, 8, 8, 8, 2);      // nt stores
#endif
          case 3: return P64(1024, 8, 2, false, 1, 8, 8, 8, 2);   // (= the lean plan the fused truncation / padding uses: no extra kernels)
        }
      case 2048:
        if (plain) return P64F(2048, 16, 2, false, 1, 8 | 3, 16, 16, 8);
        return P64(2048, 16, 2, false, 1, 16, 16, 8);
      case 4096: return P64(4096, 16, 1, false, 1, 16, 16, 16);
    }
  } else if (d.tw_hi && d.out_es == 1 && d.mode == MODE_C2C && !d.tr_dir && d.n >= 64 && variant != 9) {
    // first pass of a four-step transform: strided loads in 256-byte segments, four-step
    // twiddle, then a transposing store through LDS so that each output line is written in
    // whole rows (C2, 64 x 2^20, both passes: 1.044 -> 1.010 ms; variant 9 = the plain store for A/B)
    switch (d.n) {
      case 64: return P64F(64, 8, 16, true, 1, 32, 8, 8);
      case 128: return P64F(128, 8, 16, true, 1, 32, 8, 8, 2);
      case 256: return P64F(256, 8, 16, true, 1, 32, 8, 8, 4);
      case 512: return P64F(512, 8, 16, true, 1, 32, 8, 8, 8);
      case 1024: return P64F(1024, 16, 16, true, 4, 32, 16, 16, 4);
      case 2048: return P64F(2048, 16, 8, true, 4, 32, 16, 16, 8);
      case 4096: return P64F(4096, 16, 4, true, 4, 32, 16, 16, 16);
    }
  } else if (d.mode != MODE_C2C || d.tw_hi || d.tr_dir || d.out_es == 1 || d.in_es == 1) {
    // Strided passes that are not plain c2c column passes: r2c / c2r along a strided axis (halved
    // axis is not the array's last axis), fused truncation / padding, the second four-step pass.
    // Up to n = 1024 they now take 16 columns per tile like the plain passes (these variants
    // used to spill at R = 16 until the row index was laundered, see fft_pow2_impl.h); the lean
    // R = 8 plans with 128-byte segments remain for n >= 2048 and as variant 9.
    // R4: the complex pass with the fused TRUNCATING store (forward direction of a 3/2-rule transform) on 32 values per
    // thread / one exchange: 683^3 -> 1024^3 c128 forward 12.0-12.3 -> 11.5-11.7 ms.  The zero-padding LOAD side on the same
    // plan loses badly (backward 12.3-12.5 -> 15.9 ms) and keeps 16 values per thread (profiles/r04_variant_cols_r32.txt)
    if (d.n == 1024 && d.tr_dir == 1 && d.mode == MODE_C2C && !d.tw_hi && !d.in_lgp && !d.out_lgp && variant != 17 && variant != 9)
      return launch_pow2_one<double, 1024, 32, 16, true, true, 2, 16, MODE_C2C, false, 32, 32>(d, in, out, s);
    if (variant != 9) {
      switch (d.n) {
        case 64: return P64(64, 8, 16, true, 1, 8, 8);
        case 128: return P64(128, 8, 16, true, 1, 8, 8, 2);
        case 256: return P64(256, 8, 16, true, 1, 8, 8, 4);
        case 512: return P64(512, 8, 16, true, 1, 8, 8, 8);
        case 1024: return P64(1024, 16, 16, true, 4, 16, 16, 4);
      }
    }
    switch (d.n) {
      case 16: return P64(16, 4, 16, true, 1, 4, 4);
      case 32: return P64(32, 8, 16, true, 1, 8, 4);
      case 64: return P64(64, 8, 8, true, 1, 8, 8);
      case 128: return P64(128, 8, 8, true, 1, 8, 8, 2);
      case 256: return P64(256, 8, 8, true, 1, 8, 8, 4);
      case 512: return P64(512, 8, 8, true, 1, 8, 8, 8);
      case 1024: return P64(1024, 8, 8, true, 1, 8, 8, 8, 2);
      case 2048: return P64(2048, 8, 4, true, 1, 8, 8, 8, 4);
      case 4096: return P64(4096, 8, 2, true, 1, 8, 8, 8, 8);
    }
  } else {
    // strided axis: 16 adjacent columns = 256-byte segments wherever the thread budget allows
    // Round 6: the non-temporal streams of the n = 512 / 1024 defaults cost 20-27 % where rows do not start on 128-byte lines and the stride is near
    // ((256,1024,513) axis 1 -- a stage of a real transform on several GPUs -- 1.221 against 0.958 ms, (512,512,513) axis 1 1.324 against 1.107 ms;
    // far strides and 520-wide rows: ahead by 2-3 % as everywhere else, profiles/r06_cols_nt_probe.txt): such passes take the plain streams.
    const bool odd_rows = !strided_lines_whole(d, 16);
    const bool odd_near = odd_rows && d.in_es < 65536 && d.out_es < 65536;
    switch (d.n) {
      case 16: return P64F(16, 4, 16, true, 1, 8, 4, 4);
      case 32: return P64F(32, 8, 16, true, 1, 8, 8, 4);
      case 64: return P64F(64, 8, 16, true, 1, 8, 8, 8);
      case 128: return P64F(128, 8, 16, true, 1, 8, 8, 8, 2);
      case 256:
        // Round 6: 32 values per thread, radices 32 x 8 = ONE exchange, 32 columns = 512-byte segments on 256 threads.  Against the
        // former default (16; 8 values per thread, radices 8.8.4, 16 columns on 512 threads), plans alternating on the same arrays
        // (tools/cols_variant_probe.py, profiles/r06_cols_t32_probe.txt): (256,256,256) axis 1 0.105 -> 0.094 ms, axis 0 0.114 -> 0.102 ms,
        // (1024,256,1024) axis 1 1.704 -> 1.523 ms, (256,1024,1024) axis 0 1.799 -> 1.641 ms.  (64 columns on 512 threads: 1.450 ms on the
        // third case, slower than the default on the first; the default's radices on 32 columns: slower everywhere.)
        if (variant == 0 && d.inner % 32 == 0 && !odd_rows && 2.0 * (double)d.batch * 256 * 16 >= 268435456.0) return P64F(256, 32, 32, true, 2, 8 | 3, 32, 8);      // (non-temporal: arrays beyond the Infinity Cache)
        switch (variant) {
          default: return P64F(256, 8, 16, true, 1, 8, 8, 8, 4);
          case 21: return P64F(256, 32, 32, true, 2, 8 | 3, 32, 8);      // (the automatic choice above, whatever the array: for A/B)
        }
      case 512:
        // Round 6: on NEAR strides (the line's entries less than 2^16 elements apart: axis 1 of a 3-D array) 32 columns = 512-byte
        // segments on 512 threads, one workgroup per CU -- the tile shape that the fused pairs of unequal planes showed
        // (fft_fused_f64.hip Fused512T32): (512,512,512) axis 1 0.831 -> 0.730 ms, (256,512,512) axis 1 0.406 -> 0.387 ms,
        // (1024,512,1024) axis 1 3.190 -> 2.828 ms.  On FAR strides (axis 0) the same tile LOSES 8 %: (512,512,512) axis 0 0.800 -> 0.865 ms,
        // (512,256,512) 0.391 -> 0.423 ms -- they keep 16 columns on 256 threads, two workgroups per CU (profiles/r06_cols_t32_probe.txt).
        if (variant == 0 && d.inner % 32 == 0 && !odd_rows && d.in_es < 65536 && d.out_es < 65536) return P64F(512, 32, 32, true, 2, 8 | 3, 32, 16);
        // (... and rows that do NOT start on 128-byte lines -- 513-wide half spectra -- on near strides: the plain streams of variant 15, below)
        if (variant == 0 && odd_near) return P64F(512, 32, 16, true, 2, 8, 32, 16);
        switch (variant) {
          case 21: return P64F(512, 32, 32, true, 2, 8 | 3, 32, 16);     // (the automatic choice above, whatever the array: for A/B)
          // R4: 32 values per thread, radices 32 x 16 = ONE exchange, 256 threads on 16 columns (two workgroups per CU),
          // non-temporal loads and stores.  Against the former default (17), same box: (512,512,512) axis 1 0.85-0.88 -> 0.79 ms,
          // axis 0 0.95 -> 0.83 ms; the C3 stage (512,256,512) axis 0 0.47-0.48 -> 0.40 ms, (256,512,512) axis 1 0.42 ->
          // 0.39-0.41 ms (profiles/r04_variant_cols_r32.txt)
          default: return P64F(512, 32, 16, true, 2, 8 | 3, 32, 16);
          case 15: return P64F(512, 32, 16, true, 2, 8, 32, 16);         // ... with plain loads and stores
          case 17: return P64F(512, 8, 16, true, 1, 8, 8, 8, 8);         // 8 values per thread, radices 8.8.8, 1024 threads (rounds 1-3; real 3-D schedules)
#ifdef GFFT_VARIANTS
          case 1: return P64F(512, 8, 8, true, 1, 8, 8, 8, 8);
#endif
        }
      case 1024:
        if (variant == 0 && odd_near) return P64F(1024, 32, 16, true, 2, 8, 32, 32);
        switch (variant) {
          // R4: 32 values per thread, radices 32 x 32 = ONE exchange, 512 threads (<= 256 VGPRs: 181), non-temporal loads
          // and stores.  Against the former default (17): the stand-alone pass of the complex 3-D schedule 6.62 -> 6.45 ms
          // (profiles/r04_ab_cols_r32.txt); the C4-on-8-GPUs stages (256,1024,512) axis 1 0.87 -> 0.80-0.82 ms and
          // (1024,256,512) axis 0 0.98-1.01 -> 0.85-0.91 ms (54 -> 59-63 % of 8 TB/s; profiles/r04_variant_cols_r32.txt).
          // Under REAL 3-D schedules it loses 1.5-3 % (profiles/r04_real_pairs.txt): plan_fused3 asks for 17 there.
          default: return P64F(1024, 32, 16, true, 2, 8 | 3, 32, 32);
          case 15: return P64F(1024, 32, 16, true, 2, 8, 32, 32);     // ... with plain loads and stores
          case 17: return P64F(1024, 16, 16, true, 4, 8, 16, 16, 4);  // 16 values per thread, radices 16.16.4, 1024 threads, <= 128 VGPRs (rounds 1-3)
#ifdef GFFT_VARIANTS
          case 1: return P64F(1024, 8, 8, true, 1, 8, 8, 8, 8, 2);    // 128-B segments, 1024 threads, 86 VGPRs
          case 2: return P64F(1024, 16, 8, true, 1, 8, 16, 16, 4);    // 512 threads, ~134 VGPRs: 1 tile/CU
          case 3: return P64F(1024, 16, 8, true, 4, 8, 16, 16, 4);    // capped at 128 VGPRs: 2 tiles/CU
          case 4: return P64F(1024, 8, 4, true, 1, 8, 8, 8, 8, 2);    // 64-B segments
          case 5: return P64F(1024, 8, 8, true, 1, 4, 8, 8, 8, 2);      // access pattern only, T=8
          case 6: return P64F(1024, 8, 8, true, 1, 7, 8, 8, 8, 2);      // ... with nt loads/stores
          case 10: return P64F(1024, 16, 16, true, 4, 4, 16, 16, 4);    // access pattern only, T=16
          case 12: return P64F(1024, 16, 16, true, 4, 8, 8, 8, 8, 2);
          case 13: return P64F(1024, 16, 16, true, 4, 8 | 2, 16, 16, 4);   // non-temporal stores
          case 14: return P64F(1024, 16, 16, true, 4, 8 | 3, 16, 16, 4);   // non-temporal loads and stores
#endif
        }
      case 2048:
        // (round 6: non-temporal streams as at n = 512 / 1024 where rows start on 128-byte lines: (2048,256,512) axis 0 2.699 -> 2.300 ms, (256,2048,512)
        // axis 1 1.923 -> 1.876 ms; 513-wide rows lose with them, fft_pow2_f32.hip)
        if (variant == 0 && !odd_rows && d.inner % 8 == 0 && 2.0 * (double)d.batch * 2048 * 16 >= 268435456.0)
          return P64F(2048, 16, 8, true, 4, 8 | 3, 16, 16, 8);
        return P64F(2048, 16, 8, true, 4, 8, 16, 16, 8);
      case 4096: return P64F(4096, 16, 4, true, 4, 8, 16, 16, 16);
    }
  }
  return hipErrorInvalidValue;
}

// ---- real-to-real (DCT / DST) passes: the same kernel with the MODE_R2R load / store adapters ----
bool pow2_r2r_supported(int n) { return n >= 64 && n <= 4096 && (n & (n - 1)) == 0; }

hipError_t launch_pow2_r2r_f64(const PassDesc &d, bool cols, const void *in, void *out, hipStream_t s) {
  if (!cols) {
    switch (d.n) {
      case 64: return launch_pow2_one<double, 64, 8, 8, false, true, 1, 0, MODE_R2R, false, 8, 8>(d, in, out, s);
      case 128: return launch_pow2_one<double, 128, 8, 4, false, true, 1, 0, MODE_R2R, false, 8, 8, 2>(d, in, out, s);
      case 256: return launch_pow2_one<double, 256, 8, 8, false, true, 1, 0, MODE_R2R, false, 8, 8, 4>(d, in, out, s);
      case 512: return launch_pow2_one<double, 512, 8, 4, false, true, 1, 0, MODE_R2R, false, 8, 8, 8>(d, in, out, s);
      case 1024: return launch_pow2_one<double, 1024, 16, 4, false, true, 1, 0, MODE_R2R, false, 16, 16, 4>(d, in, out, s);
      case 2048: return launch_pow2_one<double, 2048, 16, 2, false, true, 1, 0, MODE_R2R, false, 16, 16, 8>(d, in, out, s);
      case 4096: return launch_pow2_one<double, 4096, 16, 1, false, true, 1, 0, MODE_R2R, false, 16, 16, 16>(d, in, out, s);
    }
  } else {
    switch (d.n) {
      case 64: return launch_pow2_one<double, 64, 8, 16, true, true, 1, 0, MODE_R2R, false, 8, 8>(d, in, out, s);
      case 128: return launch_pow2_one<double, 128, 8, 16, true, true, 1, 0, MODE_R2R, false, 8, 8, 2>(d, in, out, s);
      case 256: return launch_pow2_one<double, 256, 8, 16, true, true, 1, 0, MODE_R2R, false, 8, 8, 4>(d, in, out, s);
      case 512: return launch_pow2_one<double, 512, 8, 16, true, true, 1, 0, MODE_R2R, false, 8, 8, 8>(d, in, out, s);
      case 1024: return launch_pow2_one<double, 1024, 16, 16, true, true, 1, 0, MODE_R2R, false, 16, 16, 4>(d, in, out, s);
      case 2048: return launch_pow2_one<double, 2048, 8, 4, true, true, 1, 0, MODE_R2R, false, 8, 8, 8, 4>(d, in, out, s);
      case 4096: return launch_pow2_one<double, 4096, 8, 2, true, true, 1, 0, MODE_R2R, false, 8, 8, 8, 8>(d, in, out, s);
    }
  }
  return hipErrorInvalidValue;
}

}  // namespace gfft
