// Load/store adapters shared by every 1-D pass kernel: batch-index -> address, r2c / c2r views,
// inverse-by-conjugation, fused scale, fused four-step twiddle.
#pragma once
#include "gfft_internal.h"

namespace gfft {

struct ColAddr {
  int64_t in, out, mid;
};

__device__ __forceinline__ ColAddr column_address(const PassDesc &d, int64_t b) {
  int64_t i, m, o;
  if (d.inner == 1) {
    i = 0;
    if (d.mid == 1) { m = 0; o = b; } else { o = b / d.mid; m = b - o * d.mid; }
  } else {
    int64_t bm = b / d.inner;
    i = b - bm * d.inner;
    if (d.mid == 1) { m = 0; o = bm; } else { o = bm / d.mid; m = bm - o * d.mid; }
  }
  ColAddr a;
  a.in = o * d.in_os + m * d.in_ms + i * d.in_is;
  a.out = o * d.out_os + m * d.out_ms + i * d.out_is;
  a.mid = m;
  return a;
}

// logical element e (0 <= e < n) of the column whose first element is at `base`
template <typename real>
__device__ __forceinline__ cx<real> load_elem(const PassDesc &d, const void *__restrict__ in,
                                              int64_t base, int e) {
  cx<real> v;
  if (d.mode == MODE_R2C) {
    v.x = reinterpret_cast<const real *>(in)[base + (int64_t)e * d.in_es];
    v.y = 0;
  } else if (d.mode == MODE_C2R) {
    const int h = d.n >> 1;
    const bool mirror = e > h;
    const int ee = mirror ? d.n - e : e;
    v = reinterpret_cast<const cx<real> *>(in)[base + (int64_t)ee * d.in_es];
    if (mirror) v.y = -v.y;
  } else {
    v = reinterpret_cast<const cx<real> *>(in)[base + (int64_t)e * d.in_es];
  }
  if (d.conj_in) v.y = -v.y;
  return v;
}

template <typename real>
__device__ __forceinline__ cx<real> bigtwiddle(const PassDesc &d, int64_t mid, int e) {
  const int64_t x = mid * (int64_t)e;   // < big_n by construction (mid < n2, e < n1)
  const int64_t hi = x >> d.tw_L, lo = x & (((int64_t)1 << d.tw_L) - 1);
  const cx<real> a = reinterpret_cast<const cx<real> *>(d.tw_hi)[hi];
  const cx<real> b = reinterpret_cast<const cx<real> *>(d.tw_lo)[lo];
  return cmul(a, b);
}

template <typename real>
__device__ __forceinline__ void store_elem(const PassDesc &d, void *__restrict__ out, int64_t base,
                                           int e, int64_t mid, cx<real> v) {
  if (d.tw_hi) v = cmul(v, bigtwiddle<real>(d, mid, e));
  const real sc = (real)d.scale;
  v.x *= sc;
  v.y *= sc;
  if (d.conj_out) v.y = -v.y;
  if (d.mode == MODE_C2R) {
    reinterpret_cast<real *>(out)[base + (int64_t)e * d.out_es] = v.x;
  } else if (d.mode == MODE_R2C) {
    if (e <= (d.n >> 1)) reinterpret_cast<cx<real> *>(out)[base + (int64_t)e * d.out_es] = v;
  } else {
    reinterpret_cast<cx<real> *>(out)[base + (int64_t)e * d.out_es] = v;
  }
}

}  // namespace gfft
