#!/usr/bin/env python3
"""Generates csrc/fft_mixv_f64.hip / fft_mixv_f32.hip: instantiation tables of the register-resident pass for 7-smooth
lengths whose stages cannot all keep the same number of values per thread (Geo / StageV in fft_pow2_impl.h): 3 x 5 x 2^k,
7 x 2^k and their neighbours.  A plan is a radix sequence on R values per thread such that every stage's
R_s = floor(R / r_s) * r_s divides n; the first radix is not a power of two (its autosort scatter then runs in odd
multiples: no LDS slot padding needed).  The 16 lengths of the first version keep their hand-picked, measured plans."""
import os

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(os.path.dirname(HERE), 'mpi4py-fft_amd', 'csrc')
RADS = [20, 16, 15, 12, 10, 8, 7, 5, 4, 3, 2]
ALTS = set()      # lengths that carry the measured alternatives: {960, 896, 840, 672, 480, 1440} for profiles/r05_ab_mixv_variants.txt, none since

HAND = {240: (15, 16), 480: (15, 16, 2), 960: (15, 16, 4), 1920: (15, 16, 8), 3840: (15, 16, 16),
        720: (15, 3, 16), 1440: (15, 3, 16, 2), 2880: (15, 3, 16, 4), 1200: (15, 5, 16), 2400: (15, 5, 16, 2),
        112: (7, 16), 224: (7, 16, 2), 448: (7, 16, 4), 896: (7, 16, 8), 1792: (7, 16, 16), 3584: (7, 16, 16, 2)}


def smooth7(n):
    for p in (2, 3, 5, 7):
        while n % p == 0:
            n //= p
    return n == 1


def covered():
    """lengths the other tables (2^k, 3^b 2^k, 5^c 2^k) already run in one pass"""
    import re
    out = {16, 32, 64, 128, 256, 512, 1024, 2048, 4096}
    for f, fn in (('fft_mix3_f64.hip', 'mix3_supported'), ('fft_mix5_f64.hip', 'mix5_supported')):
        src = open(os.path.join(CSRC, f)).read()
        out |= {int(x) for x in re.findall(r'case (\d+):', src.split('bool ' + fn)[1].split('return true')[0])}
    return out


def stage_R(R, r):
    return (R // r) * r


def geometry(n, R, rads):
    Rs = [stage_R(R, r) for r in rads]
    if any(n % x for x in Rs):
        return None
    return Rs, max(n // x for x in Rs)


def search(n, R, max_stages=3):
    best = None

    def rec(m, seq):
        nonlocal best
        if m == 1:
            if best is None or len(seq) < len(best):
                best = list(seq)
            return
        if len(seq) >= max_stages:
            return
        for r in RADS:
            if r > R or m % r or n % stage_R(R, r):
                continue
            if not seq and (r & (r - 1)) == 0:
                continue                      # first radix: not a power of two
            rec(m // r, seq + [r])
    rec(n, [])
    return best


def plan_for(n):
    if n in HAND:
        return 16, list(HAND[n])
    for R in (16, 20, 12, 15):
        p = search(n, R)
        if p:
            return R, p
    return None


def lds_words(n, T, w4):
    NP = n + 1
    if w4:
        return (NP + 31) // 32 * 32 + (17 if T >= 32 else 32 // max(T, 1))
    return (NP + 31) // 32 * 32 + 17 if T >= 16 else ((NP + 31) // 32 * 32 + 8 if T == 4 else (NP + 15) // 16 * 16 + 2)


def pow2_floor(x):
    p = 1
    while p * 2 <= x:
        p *= 2
    return p


def configs(real_bytes):
    have = covered()
    rows, cols, half, sizes = [], [], [], []
    for n in sorted(set(HAND) | {m for m in range(96, 4097, 2) if smooth7(m) and m not in have}):
        pl = plan_for(n)
        if pl is None:
            continue
        R, rads = pl
        g = geometry(n, R, rads)
        if g is None:
            continue
        _, tpc = g
        if tpc > 1024:
            continue
        # rows: >= 64 threads, about 256
        t = max(1, pow2_floor(max(1, 256 // tpc)))
        while t * tpc < 64:
            t *= 2
        if t * tpc > 1024:
            continue
        # strided: fp64 16 columns, fp32 32 (on twice the values per thread where every stage still divides n)
        Rc, want = R, 16
        if real_bytes == 4:
            want = 32
            if geometry(n, 2 * R, rads) is not None and 2 * R <= 32:
                Rc = 2 * R
        _, tpc_c = geometry(n, Rc, rads)
        T = pow2_floor(max(1, min(want, 1024 // tpc_c)))
        while T >= 4 and T * lds_words(n, T, real_bytes == 4) * real_bytes > 160 * 1024:
            T //= 2
        if T < 4:
            continue
        while T * tpc_c < 64:
            T *= 2
        thr = T * tpc_c
        minw = 4 if thr > 512 else (2 if thr > 256 else 1)
        if Rc * 2 * (real_bytes // 4) > 64 and thr > 512:
            minw = 4
        sizes.append(n)
        r = ', '.join(map(str, rads))
        rows.append((n, R, t, r))
        # measured alternatives, selected per PLAN by option mixv_variant (tools/ab_combo_probe.py): 1 = fp64 on twice the values per
        # thread, 512-thread workgroups of up to 256 VGPRs, non-temporal loads and stores (what the 2^k tables' n = 512 / 1024 strided
        # kernels run since round 4); 2 = fp32 on the fp64 plan's values per thread (no spills, half the segment width)
        wide = None
        if n in ALTS and real_bytes == 8 and 2 * R <= 32 and geometry(n, 2 * R, rads) is not None:
            _, tpc_w = geometry(n, 2 * R, rads)
            Tw = pow2_floor(max(1, min(16, 512 // tpc_w)))
            while Tw >= 4 and Tw * lds_words(n, Tw, False) * 8 > 160 * 1024:
                Tw //= 2
            if Tw >= T and Tw * tpc_w >= 64:
                wide = ('1', 2 * R, Tw, 2 if Tw * tpc_w > 256 else 1, '8 | 3')
        if n in ALTS and real_bytes == 4 and Rc != R:
            _, tpc_n = geometry(n, R, rads)
            Tn = pow2_floor(max(1, min(32, 1024 // tpc_n)))
            while Tn >= 4 and Tn * lds_words(n, Tn, True) * 4 > 160 * 1024:
                Tn //= 2
            if Tn >= 4 and Tn * tpc_n >= 64:
                wide = ('2', R, Tn, 4 if Tn * tpc_n > 512 else (2 if Tn * tpc_n > 256 else 1), '8')
        cols.append((n, Rc, T, minw, r, wide))
    return sizes, rows, cols


HDR64 = '''// fp64 (complex128) one-pass kernels for 7-smooth lengths that are neither 2^k, 3^b 2^k nor 5^c 2^k: 3 x 5 x 2^k (240 ... 3840),
// 7 x 2^k (112 ... 3584) and their neighbours (720, 1200, 336, 560, 600, 840, 1008, 1680 ...) -- the grid sizes between the
// powers of two which rounds 1-4 ran as TWO passes per axis (960 = 48 x 20: 960^3 complex128 at 0.29 of the 2 S roofline).
// No single number of values per thread serves a radix-15 and a radix-16 stage; here every stage keeps as many as its
// radix divides -- 15 of the 16 in the radix-15 stage -- on a column of max_s n / R_s threads (Geo / StageV,
// fft_pow2_impl.h).  The first radix is never a power of two: its scatter runs in odd multiples (no LDS slot padding needed).
// Plain complex passes only (TABLE_FLAGS 8): natural layouts, no fused truncation, no four-step twiddle -- the planner keeps
// other uses of these lengths on the two-pass / generic paths -- plus plain packed-real rows of twice the length.
// The reference's own tests live on such sizes (tests/test_libfft.py:26-27, tests/test_mpifft.py:57-111).
// GENERATED by tools/gen_mixv_tables.py -- edit the generator, not the cases.
'''
HDR32 = '''// fp32 (complex64) one-pass kernels for the lengths of fft_mixv_f64.hip (see there).  A complex64 is 8 bytes, so the strided
// kernels take twice the values per thread (the radix-15 stage keeps 30 of 32) and 32 adjacent columns where every stage
// still divides the length.  GENERATED by tools/gen_mixv_tables.py -- edit the generator, not the cases.
'''
BODY = '''#include "fft_pow2_impl.h"

namespace gfft {

%(wide_decl)s#define %(X)s(N, R, T, COLS, MINW, ...) \\
  launch_pow2_inst<%(real)s, N, R, T, COLS, true, MINW, 8, __VA_ARGS__>(d, in, out, s)
// lengths divisible by 3 -- what the 3/2-rule makes of 2^k, 5 x 2^k, 7 x 2^k ... (640 -> 960, 1280 -> 1920, 448 -> 672) -- also carry
// the fused truncating store (forward) and zero-padding load (backward) of libfft.py:263-311 (FLAGS 16, 16 | 64)
#define %(X)sT(N, R, T, COLS, MINW, ...)                                                                                          \\
  (d.tr_dir == 1 ? launch_pow2_one<%(real)s, N, R, T, COLS, true, MINW, 16, MODE_C2C, false, __VA_ARGS__>(d, in, out, s)           \\
   : d.tr_dir == 2 ? launch_pow2_one<%(real)s, N, R, T, COLS, true, MINW, 16 | 64, MODE_C2C, false, __VA_ARGS__>(d, in, out, s) \\
                   : launch_pow2_inst<%(real)s, N, R, T, COLS, true, MINW, 8, __VA_ARGS__>(d, in, out, s))
%(supp)s
hipError_t launch_mixv_%(sfx)s(const PassDesc &d, bool cols, int variant, const void *in, void *out, hipStream_t s) {
  (void)variant;
  if (d.mode != MODE_C2C) return hipErrorInvalidValue;
  if (!cols) {
    switch (d.n) {      // rows: whole rows per workgroup
%(rows)s
    }
  } else {
    switch (d.n) {      // strided: adjacent columns = 256-byte segments while the tile fits 1024 threads and the LDS
%(cols)s
    }
  }
  return hipErrorInvalidValue;
}

// packed-real rows of 2 n reals (MODE_R2C_H / MODE_C2R_H, fft_real_f64.hip) on the same row plans: the Hermitian pass runs in
// the geometry of the side it sits on (after the last stage for r2c, before the first for c2r).  Plain rows only.
template <int MODE>
static hipError_t halfv_%(sfx)s(const PassDesc &d, const void *in, void *out, hipStream_t s) {
  if (d.ub_p > 1) return hipErrorInvalidValue;
  if (d.tr_dir) {      // fused truncating store (r2c) / zero-padding load (c2r): the lengths divisible by 3
    if ((MODE == MODE_R2C_H) != (d.tr_dir == 1)) return hipErrorInvalidValue;
    constexpr int TF = MODE == MODE_R2C_H ? 16 : (16 | 64);
    switch (d.n) {
%(halft)s
    }
    return hipErrorInvalidValue;
  }
  switch (d.n) {
%(half)s
  }
  return hipErrorInvalidValue;
}

hipError_t launch_real_half_mixv_%(sfx)s(const PassDesc &d, const void *in, void *out, hipStream_t s) {
  if (d.mode == MODE_R2C_H) return halfv_%(sfx)s<MODE_R2C_H>(d, in, out, s);
  if (d.mode == MODE_C2R_H) return halfv_%(sfx)s<MODE_C2R_H>(d, in, out, s);
  return hipErrorInvalidValue;
}

}  // namespace gfft
'''


def emit(real, sfx, X, real_bytes, hdr, with_supp):
    sizes, rows, cols = configs(real_bytes)
    supp = ''
    if with_supp:
        lines = []
        for i in range(0, len(sizes), 12):
            lines.append('    ' + ' '.join('case %d:' % n for n in sizes[i:i + 12]))
        supp = '\nbool mixv_supported(int n) {\n  switch (n) {\n%s\n      return true;\n  }\n  return false;\n}\n' % '\n'.join(lines)
    txt = hdr + BODY % dict(
        X=X, real=real, sfx=sfx, supp=supp,
        wide_decl='',
        rows='\n'.join('      case %d: return %s(%d, %d, %d, false, 1, %s);' % (n, X + ('T' if n % 3 == 0 else ''), n, R, t, r) for n, R, t, r in rows),
        cols='\n'.join('      case %d: %sreturn %s(%d, %d, %d, true, %d, %s);' % (
                            n, ('if (variant == %s && !d.tr_dir) return launch_pow2_inst<%s, %d, %d, %d, true, true, %d, %s, %s>(d, in, out, s);\n                '
                                % (w[0], real, n, w[1], w[2], w[3], w[4], r)) if w else '',
                            X + ('T' if n % 3 == 0 else ''), n, R, T, mw, r) for n, R, T, mw, r, w in cols),
        halft='\n'.join('      case %d: return launch_pow2_one<%s, %d, %d, %d, false, true, 1, TF, MODE, false, %s>(d, in, out, s);'
                        % (n, real, n, R, t, r) for n, R, t, r in rows if n % 3 == 0),
        half='\n'.join('    case %d: return launch_pow2_one<%s, %d, %d, %d, false, true, 1, 0, MODE, false, %s>(d, in, out, s);'
                       % (n, real, n, R, t, r) for n, R, t, r in rows))
    open(os.path.join(CSRC, 'fft_mixv_%s.hip' % sfx), 'w').write(txt)
    return sizes, rows, cols


s64, r64, c64 = emit('double', 'f64', 'X64', 8, HDR64, True)
s32, r32, c32 = emit('float', 'f32', 'X32', 4, HDR32, False)
assert s64 == s32, (sorted(set(s64) ^ set(s32)))
print(len(s64), 'lengths:', s64)
for a, b in zip(c64, c32):
    print(a, b[:5])
