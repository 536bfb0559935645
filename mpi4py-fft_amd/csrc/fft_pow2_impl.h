// Power-of-two 1-D pass for gfx950: register-resident Stockham with LDS exchange.
//
// A workgroup owns a tile of T columns of length N.  Each thread keeps R elements of one column
// in VGPRs for the whole pass (element slots e = t + q*N/R, q < R: the constant-geometry read
// side of Stockham), runs radix-16/8/4/2 butterflies on them, and between stages scatters
// through LDS (autosort write side) and reads its slots back.  First-stage inputs come straight
// from HBM, last-stage outputs go straight to HBM, so a pass reads and writes the array once.
//
//   ROWS kernel: lanes run along the transform axis (contiguous rows, 16 B/lane coalesced).
//   COLS kernel: lanes run along T adjacent columns (T*sizeof(complex) = 128 / 256 B segments),
//                the transform axis is strided: no transposes anywhere in a 3-D transform.
//
// LDS exchange is either SPLIT (real plane, then imaginary plane: halves the footprint, which is
// what lets 16 fp64 columns of N = 1024 fit the 160 KiB) or whole complex values (fp32 only).
// Twiddles: log2(r) table lookups per butterfly (w^k, w^2k, w^4k, w^8k; exact table built in
// long double on the host), the other powers by <=3 multiplications.  Inverse transforms swap
// re/im on load and store (pass_io.h), so only forward butterflies exist.
#pragma once
#include "gfft_internal.h"
#include "pass_io.h"

namespace gfft {

template <typename real> struct K {
  static constexpr real SQ = (real)0.70710678118654752440084436210485;   // cos(pi/4)
  static constexpr real C1 = (real)0.92387953251128675612818318939679;   // cos(pi/8)
  static constexpr real S1 = (real)0.38268343236508977172845998403040;   // sin(pi/8)
};

// ---- in-register radix-r DFTs (forward), natural order in and out, elements v[0], v[S], ...
template <typename real, int S> __device__ __forceinline__ void dft2(cx<real> *v) {
  cx<real> a = v[0], b = v[S];
  v[0] = a + b;
  v[S] = a - b;
}

template <typename real> __device__ __forceinline__ void bf4(cx<real> &a0, cx<real> &a1, cx<real> &a2, cx<real> &a3) {
  cx<real> t0 = a0 + a2, t1 = a0 - a2, t2 = a1 + a3, t3 = mul_mi(a1 - a3);
  a0 = t0 + t2;
  a1 = t1 + t3;
  a2 = t0 - t2;
  a3 = t1 - t3;
}

template <typename real, int S> __device__ __forceinline__ void dft4(cx<real> *v) {
  bf4(v[0], v[S], v[2 * S], v[3 * S]);
}

template <typename real> __device__ __forceinline__ cx<real> mul_w8_1(cx<real> a) {  // * exp(-i pi/4)
  return {(a.x + a.y) * K<real>::SQ, (a.y - a.x) * K<real>::SQ};
}
template <typename real> __device__ __forceinline__ cx<real> mul_w8_3(cx<real> a) {  // * exp(-3i pi/4)
  return {(a.y - a.x) * K<real>::SQ, -(a.x + a.y) * K<real>::SQ};
}
template <typename real> __device__ __forceinline__ cx<real> mul_c(cx<real> a, real wx, real wy) {
  return {a.x * wx - a.y * wy, a.x * wy + a.y * wx};
}

template <typename real, int S> __device__ __forceinline__ void dft8(cx<real> *v) {
  // x[n] = v[n*S]; n = i + 2a
  bf4(v[0], v[2 * S], v[4 * S], v[6 * S]);
  bf4(v[1 * S], v[3 * S], v[5 * S], v[7 * S]);
  v[3 * S] = mul_w8_1(v[3 * S]);  // Y_1[1]
  v[5 * S] = mul_mi(v[5 * S]);    // Y_1[2]
  v[7 * S] = mul_w8_3(v[7 * S]);  // Y_1[3]
  // radix-2 over i: (x[2k1], x[2k1+1]) -> X[k1], X[k1+4]
  cx<real> o[8];
#pragma unroll
  for (int k1 = 0; k1 < 4; ++k1) {
    cx<real> a = v[(2 * k1) * S], b = v[(2 * k1 + 1) * S];
    o[k1] = a + b;
    o[k1 + 4] = a - b;
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) v[k * S] = o[k];
}

template <typename real, int S> __device__ __forceinline__ void dft16(cx<real> *v) {
  // x[n] = v[n*S]; n = i + 4a.  Step A: radix-4 over a.
#pragma unroll
  for (int i = 0; i < 4; ++i) bf4(v[i * S], v[(i + 4) * S], v[(i + 8) * S], v[(i + 12) * S]);
  // twiddles W16^(i*k1) on x[i + 4*k1]
  const real C1 = K<real>::C1, S1 = K<real>::S1, SQ = K<real>::SQ;
  v[5 * S] = mul_c(v[5 * S], C1, -S1);    // i=1,k1=1: W^1
  v[9 * S] = mul_w8_1(v[9 * S]);          // i=1,k1=2: W^2
  v[13 * S] = mul_c(v[13 * S], S1, -C1);  // i=1,k1=3: W^3
  v[6 * S] = mul_w8_1(v[6 * S]);          // i=2,k1=1: W^2
  v[10 * S] = mul_mi(v[10 * S]);          // i=2,k1=2: W^4
  v[14 * S] = mul_w8_3(v[14 * S]);        // i=2,k1=3: W^6
  v[7 * S] = mul_c(v[7 * S], S1, -C1);    // i=3,k1=1: W^3
  v[11 * S] = mul_w8_3(v[11 * S]);        // i=3,k1=2: W^6
  v[15 * S] = mul_c(v[15 * S], -C1, S1);  // i=3,k1=3: W^9
  (void)SQ;
  // Step B: radix-4 over i on (x[4k1..4k1+3]) -> X[k1 + 4*k2] at x[4k1 + k2]
#pragma unroll
  for (int k1 = 0; k1 < 4; ++k1) bf4(v[(4 * k1) * S], v[(4 * k1 + 1) * S], v[(4 * k1 + 2) * S], v[(4 * k1 + 3) * S]);
  // 4x4 transpose to natural order
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = a + 1; b < 4; ++b) {
      cx<real> t = v[(4 * a + b) * S];
      v[(4 * a + b) * S] = v[(4 * b + a) * S];
      v[(4 * b + a) * S] = t;
    }
}

// compile-time cos of 2 pi m / 32 for the radix-32 internal twiddles
template <typename real> struct W32 {
  static constexpr real C[32] = {
      (real)1.0, (real)0.98078528040323044913, (real)0.92387953251128675613, (real)0.83146961230254523708,
      (real)0.70710678118654752440, (real)0.55557023301960222474, (real)0.38268343236508977173, (real)0.19509032201612826785,
      (real)0.0, (real)-0.19509032201612826785, (real)-0.38268343236508977173, (real)-0.55557023301960222474,
      (real)-0.70710678118654752440, (real)-0.83146961230254523708, (real)-0.92387953251128675613, (real)-0.98078528040323044913,
      (real)-1.0, (real)-0.98078528040323044913, (real)-0.92387953251128675613, (real)-0.83146961230254523708,
      (real)-0.70710678118654752440, (real)-0.55557023301960222474, (real)-0.38268343236508977173, (real)-0.19509032201612826785,
      (real)0.0, (real)0.19509032201612826785, (real)0.38268343236508977173, (real)0.55557023301960222474,
      (real)0.70710678118654752440, (real)0.83146961230254523708, (real)0.92387953251128675613, (real)0.98078528040323044913};
  // exp(-2 pi i m / 32) = (C[m], -C[(m + 24) % 32])   since sin(x) = cos(x - pi/2)
  static __device__ __forceinline__ cx<real> mul(cx<real> a, int m) {
    const real wx = C[m % 32], wy = -C[(m + 24) % 32];
    return {a.x * wx - a.y * wy, a.x * wy + a.y * wx};
  }
};

template <typename real, int S> __device__ __forceinline__ void dft32(cx<real> *v) {
  // n = i + 4a (i < 4, a < 8);  X[k1 + 8 k2] = sum_i W4^(i k2) W32^(i k1) sum_a x[i + 4a] W8^(a k1)
#pragma unroll
  for (int i = 0; i < 4; ++i) dft8<real, 4 * S>(v + i * S);          // Y_i[k1] at position i + 4 k1
#pragma unroll
  for (int i = 1; i < 4; ++i)
#pragma unroll
    for (int k1 = 1; k1 < 8; ++k1) {
      if (i * k1 == 4) v[(i + 4 * k1) * S] = mul_w8_1(v[(i + 4 * k1) * S]);
      else if (i * k1 == 8) v[(i + 4 * k1) * S] = mul_mi(v[(i + 4 * k1) * S]);
      else if (i * k1 == 12) v[(i + 4 * k1) * S] = mul_w8_3(v[(i + 4 * k1) * S]);
      else v[(i + 4 * k1) * S] = W32<real>::mul(v[(i + 4 * k1) * S], i * k1);
    }
#pragma unroll
  for (int k1 = 0; k1 < 8; ++k1) bf4(v[(4 * k1) * S], v[(4 * k1 + 1) * S], v[(4 * k1 + 2) * S], v[(4 * k1 + 3) * S]);
  cx<real> o[32];
#pragma unroll
  for (int k1 = 0; k1 < 8; ++k1)
#pragma unroll
    for (int k2 = 0; k2 < 4; ++k2) o[k1 + 8 * k2] = v[(4 * k1 + k2) * S];
#pragma unroll
  for (int k = 0; k < 32; ++k) v[k * S] = o[k];
}

// compile-time exp(-2 pi i m / 64) for the radix-64 internal twiddles (Taylor evaluation in long double, namespace ct below
// is declared later: a small local copy of the two series keeps this table self-contained)
namespace ct64 {
constexpr long double PI = 3.14159265358979323846264338327950288L;
constexpr long double sin_s(long double x) { long double t = x, s = x; for (int k = 1; k < 16; ++k) { t *= -x * x / ((2 * k) * (2 * k + 1)); s += t; } return s; }
constexpr long double cos_s(long double x) { long double t = 1, s = 1; for (int k = 1; k < 16; ++k) { t *= -x * x / ((2 * k - 1) * (2 * k)); s += t; } return s; }
// angles 0 .. 2 pi (49/64 at most here), reduced to |x| <= pi/4 by quadrant
constexpr long double cosq(int m) {          // cos(2 pi m / 64), m in 0..63
  const int mm = m % 64;
  const int oct = mm / 8, r = mm % 8;        // angle = oct * pi/4 + r * pi/32
  const long double a = PI * r / 32;
  const long double c = cos_s(a), s = sin_s(a);
  const long double h = 0.70710678118654752440084436210485L;
  // cos(oct pi/4 + a) = cos(oct pi/4) cos a - sin(oct pi/4) sin a
  const long double co[8] = {1, h, 0, -h, -1, -h, 0, h}, so[8] = {0, h, 1, h, 0, -h, -1, -h};
  return co[oct] * c - so[oct] * s;
}
constexpr long double sinq(int m) {
  const int mm = m % 64;
  const int oct = mm / 8, r = mm % 8;
  const long double a = PI * r / 32;
  const long double c = cos_s(a), s = sin_s(a);
  const long double h = 0.70710678118654752440084436210485L;
  const long double co[8] = {1, h, 0, -h, -1, -h, 0, h}, so[8] = {0, h, 1, h, 0, -h, -1, -h};
  return so[oct] * c + co[oct] * s;
}
}  // namespace ct64
template <typename real> struct W64 {
  // a * exp(-2 pi i m / 64), m a compile-time value after unrolling
  static __device__ __forceinline__ cx<real> mul(cx<real> a, int m) {
    constexpr struct Tab { real c[64], s[64]; constexpr Tab() : c(), s() { for (int k = 0; k < 64; ++k) { c[k] = (real)ct64::cosq(k); s[k] = (real)-ct64::sinq(k); } } } tab{};
    const real wx = tab.c[m % 64], wy = tab.s[m % 64];
    return {a.x * wx - a.y * wy, a.x * wy + a.y * wx};
  }
};

template <typename real, int S> __device__ __forceinline__ void dft64(cx<real> *v) {
  // n = i + 8a (i < 8, a < 8);  X[k1 + 8 k2] = sum_i W8^(i k2) W64^(i k1) sum_a x[i + 8a] W8^(a k1)
#pragma unroll
  for (int i = 0; i < 8; ++i) dft8<real, 8 * S>(v + i * S);          // Y_i[k1] at position i + 8 k1
#pragma unroll
  for (int i = 1; i < 8; ++i)
#pragma unroll
    for (int k1 = 1; k1 < 8; ++k1) {
      const int m = i * k1;
      if (m == 8) v[(i + 8 * k1) * S] = mul_w8_1(v[(i + 8 * k1) * S]);
      else if (m == 16) v[(i + 8 * k1) * S] = mul_mi(v[(i + 8 * k1) * S]);
      else if (m == 24) v[(i + 8 * k1) * S] = mul_w8_3(v[(i + 8 * k1) * S]);
      else v[(i + 8 * k1) * S] = W64<real>::mul(v[(i + 8 * k1) * S], m);
    }
#pragma unroll
  for (int k1 = 0; k1 < 8; ++k1) dft8<real, S>(v + 8 * k1 * S);      // over i: X[k1 + 8 k2] at position 8 k1 + k2
  cx<real> o[64];
#pragma unroll
  for (int k1 = 0; k1 < 8; ++k1)
#pragma unroll
    for (int k2 = 0; k2 < 8; ++k2) o[k1 + 8 * k2] = v[(8 * k1 + k2) * S];
#pragma unroll
  for (int k = 0; k < 64; ++k) v[k * S] = o[k];
}

template <typename real> __device__ __forceinline__ void bf3(cx<real> &a, cx<real> &b, cx<real> &c) {
  const real h = (real)0.86602540378443864676372317075294;   // sin(pi/3)
  cx<real> t1 = b + c;
  cx<real> t2 = {a.x - (real)0.5 * t1.x, a.y - (real)0.5 * t1.y};
  cx<real> t3 = {(b.x - c.x) * h, (b.y - c.y) * h};
  a = a + t1;
  b = {t2.x + t3.y, t2.y - t3.x};      // t2 - i*t3
  c = {t2.x - t3.y, t2.y + t3.x};      // t2 + i*t3
}

template <typename real, int S> __device__ __forceinline__ void dft3(cx<real> *v) { bf3(v[0], v[S], v[2 * S]); }

template <typename real, int S> __device__ __forceinline__ void dft12(cx<real> *v) {
  // x[n] = v[n*S], n = i + 3a (i < 3, a < 4); X[k1 + 4*k2] = sum_i W3^(i k2) W12^(i k1) sum_a x[i+3a] W4^(a k1)
#pragma unroll
  for (int i = 0; i < 3; ++i) bf4(v[i * S], v[(i + 3) * S], v[(i + 6) * S], v[(i + 9) * S]);   // Y_i[k1] at i + 3*k1
  const real h = (real)0.86602540378443864676372317075294;
  v[4 * S] = mul_c(v[4 * S], h, (real)-0.5);             // i=1,k1=1: W12^1
  v[7 * S] = mul_c(v[7 * S], (real)0.5, -h);             // i=1,k1=2: W12^2
  v[10 * S] = mul_mi(v[10 * S]);                         // i=1,k1=3: W12^3 = -i
  v[5 * S] = mul_c(v[5 * S], (real)0.5, -h);             // i=2,k1=1: W12^2
  v[8 * S] = mul_c(v[8 * S], (real)-0.5, -h);            // i=2,k1=2: W12^4
  v[11 * S] = {-v[11 * S].x, -v[11 * S].y};              // i=2,k1=3: W12^6 = -1
#pragma unroll
  for (int k1 = 0; k1 < 4; ++k1) bf3(v[(3 * k1) * S], v[(3 * k1 + 1) * S], v[(3 * k1 + 2) * S]);  // X[k1+4k2] at 3*k1 + k2
  cx<real> o[12];
#pragma unroll
  for (int k1 = 0; k1 < 4; ++k1)
#pragma unroll
    for (int k2 = 0; k2 < 3; ++k2) o[k1 + 4 * k2] = v[(3 * k1 + k2) * S];
#pragma unroll
  for (int k = 0; k < 12; ++k) v[k * S] = o[k];
}

template <typename real>
__device__ __forceinline__ void bf5(cx<real> &x0, cx<real> &x1, cx<real> &x2, cx<real> &x3, cx<real> &x4) {
  const real c1 = (real)0.30901699437494742410229341718282;    // cos(2 pi/5)
  const real c2 = (real)-0.80901699437494742410229341718282;   // cos(4 pi/5)
  const real s1 = (real)0.95105651629515357211643933337938;    // sin(2 pi/5)
  const real s2 = (real)0.58778525229247312916870595463907;    // sin(4 pi/5)
  const cx<real> t1 = x1 + x4, t2 = x2 + x3, t3 = x1 - x4, t4 = x2 - x3;
  const cx<real> a = {x0.x + c1 * t1.x + c2 * t2.x, x0.y + c1 * t1.y + c2 * t2.y};
  const cx<real> b = {x0.x + c2 * t1.x + c1 * t2.x, x0.y + c2 * t1.y + c1 * t2.y};
  const cx<real> s = {s1 * t3.x + s2 * t4.x, s1 * t3.y + s2 * t4.y};
  const cx<real> u = {s2 * t3.x - s1 * t4.x, s2 * t3.y - s1 * t4.y};
  x0 = x0 + t1 + t2;
  x1 = {a.x + s.y, a.y - s.x};     // a - i s
  x4 = {a.x - s.y, a.y + s.x};     // a + i s
  x2 = {b.x + u.y, b.y - u.x};     // b - i u
  x3 = {b.x - u.y, b.y + u.x};     // b + i u
}

template <typename real, int S> __device__ __forceinline__ void dft5(cx<real> *v) {
  bf5(v[0], v[S], v[2 * S], v[3 * S], v[4 * S]);
}

// compile-time cos / sin of 2 pi m / 20 for the radix-10/20 internal twiddles
template <typename real> struct W20 {
  static constexpr real C[20] = {
      (real)1.0, (real)0.95105651629515357212, (real)0.80901699437494742410, (real)0.58778525229247312917,
      (real)0.30901699437494742410, (real)0.0, (real)-0.30901699437494742410, (real)-0.58778525229247312917,
      (real)-0.80901699437494742410, (real)-0.95105651629515357212, (real)-1.0, (real)-0.95105651629515357212,
      (real)-0.80901699437494742410, (real)-0.58778525229247312917, (real)-0.30901699437494742410, (real)0.0,
      (real)0.30901699437494742410, (real)0.58778525229247312917, (real)0.80901699437494742410,
      (real)0.95105651629515357212};
  // exp(-2 pi i m / 20) = (C[m], -C[(m + 15) % 20])   since sin(x) = cos(x - pi/2)
  static __device__ __forceinline__ cx<real> mul(cx<real> a, int m) {
    const real wx = C[m % 20], wy = -C[(m + 15) % 20];
    return {a.x * wx - a.y * wy, a.x * wy + a.y * wx};
  }
};

template <typename real, int S> __device__ __forceinline__ void dft10(cx<real> *v) {
  // n = i + 2a (i < 2, a < 5);  X[k1 + 5 k2] = sum_i (-1)^(i k2) W10^(i k1) sum_a x[i + 2a] W5^(a k1)
  bf5(v[0], v[2 * S], v[4 * S], v[6 * S], v[8 * S]);
  bf5(v[1 * S], v[3 * S], v[5 * S], v[7 * S], v[9 * S]);
#pragma unroll
  for (int k1 = 1; k1 < 5; ++k1) v[(1 + 2 * k1) * S] = W20<real>::mul(v[(1 + 2 * k1) * S], 2 * k1);   // W10^k1
  cx<real> o[10];
#pragma unroll
  for (int k1 = 0; k1 < 5; ++k1) {
    const cx<real> a = v[(2 * k1) * S], b = v[(2 * k1 + 1) * S];
    o[k1] = a + b;
    o[k1 + 5] = a - b;
  }
#pragma unroll
  for (int k = 0; k < 10; ++k) v[k * S] = o[k];
}

template <typename real, int S> __device__ __forceinline__ void dft20(cx<real> *v) {
  // n = i + 4a (i < 4, a < 5);  X[k1 + 5 k2] = sum_i W4^(i k2) W20^(i k1) sum_a x[i + 4a] W5^(a k1)
#pragma unroll
  for (int i = 0; i < 4; ++i) bf5(v[i * S], v[(i + 4) * S], v[(i + 8) * S], v[(i + 12) * S], v[(i + 16) * S]);
#pragma unroll
  for (int i = 1; i < 4; ++i)
#pragma unroll
    for (int k1 = 1; k1 < 5; ++k1) v[(i + 4 * k1) * S] = W20<real>::mul(v[(i + 4 * k1) * S], i * k1);
#pragma unroll
  for (int k1 = 0; k1 < 5; ++k1) bf4(v[(4 * k1) * S], v[(4 * k1 + 1) * S], v[(4 * k1 + 2) * S], v[(4 * k1 + 3) * S]);
  cx<real> o[20];
#pragma unroll
  for (int k1 = 0; k1 < 5; ++k1)
#pragma unroll
    for (int k2 = 0; k2 < 4; ++k2) o[k1 + 5 * k2] = v[(4 * k1 + k2) * S];
#pragma unroll
  for (int k = 0; k < 20; ++k) v[k * S] = o[k];
}

// radix 7, symmetric form: X[k] = a_k - i b_k, X[7-k] = a_k + i b_k with a_k = x0 + sum_j cos(2 pi j k / 7) (x_j + x_{7-j}),
// b_k = sum_j sin(2 pi j k / 7) (x_j - x_{7-j}), j, k = 1..3
template <typename real, int S> __device__ __forceinline__ void dft7(cx<real> *v) {
  constexpr real C1 = (real)0.62348980185873353052500488400424, C2 = (real)-0.22252093395631440428890256449679,
                 C3 = (real)-0.90096886790241912623610231950745;      // cos(2 pi k / 7)
  constexpr real S1 = (real)0.78183148246802980870844452667406, S2 = (real)0.97492791218182360701813168299393,
                 S3 = (real)0.43388373911755812047576833284836;       // sin(2 pi k / 7)
  const cx<real> x0 = v[0];
  const cx<real> p1 = v[1 * S] + v[6 * S], p2 = v[2 * S] + v[5 * S], p3 = v[3 * S] + v[4 * S];
  const cx<real> m1 = v[1 * S] - v[6 * S], m2 = v[2 * S] - v[5 * S], m3 = v[3 * S] - v[4 * S];
  v[0] = x0 + p1 + p2 + p3;
  // (cos / sin of 2 pi j k / 7 for k = 1, 2, 3: rows of the index table j k mod 7 -> +-{1, 2, 3})
  const cx<real> a1 = {x0.x + C1 * p1.x + C2 * p2.x + C3 * p3.x, x0.y + C1 * p1.y + C2 * p2.y + C3 * p3.y};
  const cx<real> a2 = {x0.x + C2 * p1.x + C3 * p2.x + C1 * p3.x, x0.y + C2 * p1.y + C3 * p2.y + C1 * p3.y};
  const cx<real> a3 = {x0.x + C3 * p1.x + C1 * p2.x + C2 * p3.x, x0.y + C3 * p1.y + C1 * p2.y + C2 * p3.y};
  const cx<real> b1 = {S1 * m1.x + S2 * m2.x + S3 * m3.x, S1 * m1.y + S2 * m2.y + S3 * m3.y};
  const cx<real> b2 = {S2 * m1.x - S3 * m2.x - S1 * m3.x, S2 * m1.y - S3 * m2.y - S1 * m3.y};
  const cx<real> b3 = {S3 * m1.x - S1 * m2.x + S2 * m3.x, S3 * m1.y - S1 * m2.y + S2 * m3.y};
  // a - i b = (a.x + b.y, a.y - b.x);  a + i b = (a.x - b.y, a.y + b.x)
  v[1 * S] = {a1.x + b1.y, a1.y - b1.x};
  v[6 * S] = {a1.x - b1.y, a1.y + b1.x};
  v[2 * S] = {a2.x + b2.y, a2.y - b2.x};
  v[5 * S] = {a2.x - b2.y, a2.y + b2.x};
  v[3 * S] = {a3.x + b3.y, a3.y - b3.x};
  v[4 * S] = {a3.x - b3.y, a3.y + b3.x};
}

// radix 15 = 3 x 5 by the prime-factor map (3 and 5 coprime: no twiddles inside): input n = (5 n1 + 3 n2) mod 15, output
// k = (10 k1 + 6 k2) mod 15, W15^(n k) = W3^(n1 k1) W5^(n2 k2); the index maps are compile-time register renamings
template <typename real, int S> __device__ __forceinline__ void dft15(cx<real> *v) {
  cx<real> a[15];
#pragma unroll
  for (int n1 = 0; n1 < 3; ++n1)
#pragma unroll
    for (int n2 = 0; n2 < 5; ++n2) a[n1 * 5 + n2] = v[((5 * n1 + 3 * n2) % 15) * S];
#pragma unroll
  for (int n1 = 0; n1 < 3; ++n1) dft5<real, 1>(a + n1 * 5);
#pragma unroll
  for (int k2 = 0; k2 < 5; ++k2) dft3<real, 5>(a + k2);
#pragma unroll
  for (int k1 = 0; k1 < 3; ++k1)
#pragma unroll
    for (int k2 = 0; k2 < 5; ++k2) v[((10 * k1 + 6 * k2) % 15) * S] = a[k1 * 5 + k2];
}

template <typename real, int r, int S> __device__ __forceinline__ void dft(cx<real> *v) {
  static_assert(r == 2 || r == 3 || r == 4 || r == 5 || r == 7 || r == 8 || r == 10 || r == 12 || r == 15 || r == 16 || r == 20 || r == 32 || r == 64, "radix");
  if constexpr (r == 15) { dft15<real, S>(v); return; }
  if constexpr (r == 7) { dft7<real, S>(v); return; }
  if constexpr (r == 2) dft2<real, S>(v);
  else if constexpr (r == 3) dft3<real, S>(v);
  else if constexpr (r == 4) dft4<real, S>(v);
  else if constexpr (r == 5) dft5<real, S>(v);
  else if constexpr (r == 8) dft8<real, S>(v);
  else if constexpr (r == 10) dft10<real, S>(v);
  else if constexpr (r == 12) dft12<real, S>(v);
  else if constexpr (r == 16) dft16<real, S>(v);
  else if constexpr (r == 32) dft32<real, S>(v);
  else if constexpr (r == 64) dft64<real, S>(v);
  else dft20<real, S>(v);
}

// ---- LDS layout -----------------------------------------------------------------------------
__host__ __device__ constexpr bool is_pow2_c(int n) { return n > 0 && (n & (n - 1)) == 0; }
// slot padding: power-of-two sizes insert one word per 16 (breaks the power-of-two scatter
// strides); sizes with a factor 3 scatter with odd multiples and need none.  PAD is a
// compile-time property of N so that pad(a + b) splits into pad(a) + constant (see Stage::run).
template <int PADSH> __host__ __device__ constexpr int pad_slot(int e) { return PADSH ? e + (e >> PADSH) : e; }
// ... one word per 32 where a ROW pass opens with a radix-32 stage: its threads scatter 32 slots apart, and with one pad
// word per 16 that is 34 words = 68 dwords == 4 (mod 32 banks) between neighbouring lanes, a 2-way conflict in every
// 16-lane group of ds_write_b64 (the fused launch on 32 values per thread: 19 % of its LDS cycles,
// profiles/r04_bench_1024c128_pmc_lds.txt); 33 words = 66 dwords == 2 is conflict free.  Strided passes run their lanes
// along the adjacent columns first (column stride CS, below) and keep the one-per-16 rule they were measured with.
template <int... RADS> struct FirstRadix { static constexpr int value = 0; };
template <int r0, int... REST> struct FirstRadix<r0, REST...> { static constexpr int value = r0; };
template <int N, int R0, bool COLS> __host__ __device__ constexpr int pad_shift() { return is_pow2_c(N) ? ((R0 == 32 && !COLS) ? 5 : 4) : 0; }

template <int N, bool COLS, int T, bool W4 = false, int R0 = 16> struct Lds {
  static constexpr int PADSH = pad_shift<N, R0, COLS>();
  static constexpr int NP = PADSH ? N + (N >> PADSH) + 1 : N + 1;
  // COLS: lanes run over T adjacent columns first, so the column stride CS (in words) decides the
  // banks.  8-byte words (fp64 planes, unsplit fp32 pairs; ds_*_b64, 64 banks):
  // T <= 8: CS == 2 (mod 16): 8 columns x 2 rows cover the 32 write banks.
  // T >= 16: CS == 17 (mod 32): 16 columns land on 16 distinct bank pairs for ds_write_b64
  // (34c mod 32 = 2c) and, with two rows per 32-lane ds_read_b64 group, on 32 distinct pairs of
  // the 64 read banks.  (With CS == 2 mod 16 the T = 16 kernel measured 48 % of its LDS cycles
  // as bank conflicts: profiles/r01b_*_pmc_lds.txt.)
  // 4-byte words (W4: fp32 planes; ds_*_b32, 32 banks, 32 lanes per LDS cycle = T columns x 32/T
  // consecutive rows): the columns take every (32/T)-th bank, CS == 32/T (mod 32), and the rows
  // of a group fill the gaps -- consecutive rows sit 1 slot apart on the read side and in the
  // stages with Ns > 1, pad(16) = 17 slots apart in a first radix-16 stage: both run through all
  // residues mod 32/T.  (The 8-byte rule on 4-byte words made row t + 1 of column c collide with
  // row t of column c + 1 in that first stage: fp32 n = 2048, T = 16 measured 48 % of its LDS
  // cycles as conflicts, profiles/r02_c5cols_pmc_lds.txt.)
  static constexpr int CS8 = T >= 16 ? ((NP + 31) / 32) * 32 + 17 : (T == 4 ? ((NP + 31) / 32) * 32 + 8 : ((NP + 15) / 16) * 16 + 2);
  static constexpr int CS4 = T >= 32 ? ((NP + 31) / 32) * 32 + 17 : ((NP + 31) / 32) * 32 + 32 / (T < 1 ? 1 : T);
  static constexpr int CS = !COLS ? NP : (W4 ? CS4 : CS8);
};

// multiply v[i + m*S] (m = 1..r-1) by w^(m*k); w = exp(-2 pi i/(Ns*r)); table stride = N/(Ns*r)
template <typename real, int N, int r, int Ns, int S>
__device__ __forceinline__ void twiddle(cx<real> *v, int k, const cx<real> *__restrict__ tw) {
  constexpr int step = N / (Ns * r);
  const cx<real> w1 = tw[k * step];
  v[1 * S] = cmul(v[1 * S], w1);
  if constexpr (r == 3) {
    v[2 * S] = cmul(v[2 * S], tw[2 * k * step]);
  } else if constexpr (r == 7) {
    const cx<real> w2 = tw[2 * k * step], w4 = tw[4 * k * step];
    v[2 * S] = cmul(v[2 * S], w2);
    v[3 * S] = cmul(v[3 * S], cmul(w1, w2));
    v[4 * S] = cmul(v[4 * S], w4);
    v[5 * S] = cmul(v[5 * S], cmul(w1, w4));
    v[6 * S] = cmul(v[6 * S], cmul(w2, w4));
  } else if constexpr (r == 15) {
    const cx<real> w2 = tw[2 * k * step], w4 = tw[4 * k * step], w8 = tw[8 * k * step];
    const cx<real> w3 = cmul(w1, w2), w12 = cmul(w4, w8);
    v[2 * S] = cmul(v[2 * S], w2);
    v[3 * S] = cmul(v[3 * S], w3);
    v[4 * S] = cmul(v[4 * S], w4);
    v[5 * S] = cmul(v[5 * S], cmul(w1, w4));
    v[6 * S] = cmul(v[6 * S], cmul(w2, w4));
    v[7 * S] = cmul(v[7 * S], cmul(w3, w4));
    v[8 * S] = cmul(v[8 * S], w8);
    v[9 * S] = cmul(v[9 * S], cmul(w1, w8));
    v[10 * S] = cmul(v[10 * S], cmul(w2, w8));
    v[11 * S] = cmul(v[11 * S], cmul(w3, w8));
    v[12 * S] = cmul(v[12 * S], w12);
    v[13 * S] = cmul(v[13 * S], cmul(w1, w12));
    v[14 * S] = cmul(v[14 * S], cmul(w2, w12));
  } else if constexpr (r == 5 || r == 10 || r == 20) {
    // binary bases from the table, the rest composed: w^m = prod over set bits of m
    const cx<real> w2 = tw[2 * k * step], w4 = tw[4 * k * step];
    const cx<real> w3 = cmul(w1, w2);
    v[2 * S] = cmul(v[2 * S], w2);
    v[3 * S] = cmul(v[3 * S], w3);
    v[4 * S] = cmul(v[4 * S], w4);
    if constexpr (r >= 10) {
      const cx<real> w8 = tw[8 * k * step];
      v[5 * S] = cmul(v[5 * S], cmul(w1, w4));
      v[6 * S] = cmul(v[6 * S], cmul(w2, w4));
      v[7 * S] = cmul(v[7 * S], cmul(w3, w4));
      v[8 * S] = cmul(v[8 * S], w8);
      v[9 * S] = cmul(v[9 * S], cmul(w1, w8));
      if constexpr (r == 20) {
        const cx<real> w16 = tw[16 * k * step];
        const cx<real> w12 = cmul(w4, w8);
        v[10 * S] = cmul(v[10 * S], cmul(w2, w8));
        v[11 * S] = cmul(v[11 * S], cmul(w3, w8));
        v[12 * S] = cmul(v[12 * S], w12);
        v[13 * S] = cmul(v[13 * S], cmul(w1, w12));
        v[14 * S] = cmul(v[14 * S], cmul(w2, w12));
        v[15 * S] = cmul(v[15 * S], cmul(w3, w12));
        v[16 * S] = cmul(v[16 * S], w16);
        v[17 * S] = cmul(v[17 * S], cmul(w1, w16));
        v[18 * S] = cmul(v[18 * S], cmul(w2, w16));
        v[19 * S] = cmul(v[19 * S], cmul(w3, w16));
      }
    }
  } else if constexpr (r == 12) {
    const cx<real> w2 = tw[2 * k * step], w4 = tw[4 * k * step], w8 = tw[8 * k * step];
    const cx<real> w3 = cmul(w1, w2);
    v[2 * S] = cmul(v[2 * S], w2);
    v[3 * S] = cmul(v[3 * S], w3);
    v[4 * S] = cmul(v[4 * S], w4);
    v[5 * S] = cmul(v[5 * S], cmul(w1, w4));
    v[6 * S] = cmul(v[6 * S], cmul(w2, w4));
    v[7 * S] = cmul(v[7 * S], cmul(w3, w4));
    v[8 * S] = cmul(v[8 * S], w8);
    v[9 * S] = cmul(v[9 * S], cmul(w1, w8));
    v[10 * S] = cmul(v[10 * S], cmul(w2, w8));
    v[11 * S] = cmul(v[11 * S], cmul(w3, w8));
  } else if constexpr (r >= 4) {
    const cx<real> w2 = tw[2 * k * step];
    const cx<real> w3 = cmul(w1, w2);
    v[2 * S] = cmul(v[2 * S], w2);
    v[3 * S] = cmul(v[3 * S], w3);
    if constexpr (r >= 8) {
      const cx<real> w4 = tw[4 * k * step];
      const cx<real> w5 = cmul(w1, w4), w6 = cmul(w2, w4), w7 = cmul(w3, w4);
      v[4 * S] = cmul(v[4 * S], w4);
      v[5 * S] = cmul(v[5 * S], w5);
      v[6 * S] = cmul(v[6 * S], w6);
      v[7 * S] = cmul(v[7 * S], w7);
      if constexpr (r >= 16) {
        const cx<real> w8 = tw[8 * k * step];
        v[8 * S] = cmul(v[8 * S], w8);
        v[9 * S] = cmul(v[9 * S], cmul(w1, w8));
        v[10 * S] = cmul(v[10 * S], cmul(w2, w8));
        v[11 * S] = cmul(v[11 * S], cmul(w3, w8));
        v[12 * S] = cmul(v[12 * S], cmul(w4, w8));
        v[13 * S] = cmul(v[13 * S], cmul(w5, w8));
        v[14 * S] = cmul(v[14 * S], cmul(w6, w8));
        v[15 * S] = cmul(v[15 * S], cmul(w7, w8));
        if constexpr (r >= 32) {
          const cx<real> w16 = tw[16 * k * step];
          const cx<real> w24 = cmul(w8, w16);
          v[16 * S] = cmul(v[16 * S], w16);
          v[17 * S] = cmul(v[17 * S], cmul(w1, w16));
          v[18 * S] = cmul(v[18 * S], cmul(w2, w16));
          v[19 * S] = cmul(v[19 * S], cmul(w3, w16));
          v[20 * S] = cmul(v[20 * S], cmul(w4, w16));
          v[21 * S] = cmul(v[21 * S], cmul(w5, w16));
          v[22 * S] = cmul(v[22 * S], cmul(w6, w16));
          v[23 * S] = cmul(v[23 * S], cmul(w7, w16));
          v[24 * S] = cmul(v[24 * S], w24);
          v[25 * S] = cmul(v[25 * S], cmul(w1, w24));
          v[26 * S] = cmul(v[26 * S], cmul(w2, w24));
          v[27 * S] = cmul(v[27 * S], cmul(w3, w24));
          v[28 * S] = cmul(v[28 * S], cmul(w4, w24));
          v[29 * S] = cmul(v[29 * S], cmul(w5, w24));
          v[30 * S] = cmul(v[30 * S], cmul(w6, w24));
          v[31 * S] = cmul(v[31 * S], cmul(w7, w24));
        }
      }
    }
  }
}

// ---- one Stockham stage + exchange, recursing over the radix list -------------------------
// (trace builds, GFFT_FUSE2_TRACE: thread 0 of a fused launch's workgroup stamps the wall clock at phase
// boundaries inside the tile as well -- see GFFT_TRACE_STAMP below, tools/fused2_trace.py)
#ifdef GFFT_FUSE2_TRACE
static __shared__ unsigned long long *gfft_trace_slot;
#define GFFT_PHASE(i)                                                                  \
  if (__builtin_amdgcn_readfirstlane(threadIdx.x) == 0) {                               \
    if (threadIdx.x == 0 && gfft_trace_slot) gfft_trace_slot[i] = wall_clock64();      \
  }
#else
#define GFFT_PHASE(i)
#endif

// Synchronisation between the write and the read side of an LDS exchange.  A row pass whose rows are at most
// one wavefront wide (N / R threads per row, 64 % (N / R) == 0) keeps every row -- its registers AND its LDS
// region -- inside ONE wave: DS instructions of a wave execute in order, so the exchange needs no s_barrier
// at all, only the compiler kept from reordering (WL = true).  The waves of the workgroup then drift apart
// and one wave's loads / stores overlap another's butterflies and exchanges -- which the lock step of 14
// barriers per tile forbade (tools/fused2_trace.py: 2 x 4-5 us of a 22 us tile sat in the exchanges).
template <bool WL> __device__ __forceinline__ void tile_sync() {
  if constexpr (WL) {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  } else {
    __syncthreads();
  }
}
// (Strided passes -- lanes along T adjacent columns, a column spread over all waves -- keep their barriers.
// Their LAST exchange alone can be made wave local -- it only moves values inside groups of r_last threads
// N / R / r_last apart, so with a wave's 64 / T row-threads chosen that far apart three of its four barriers go
// -- and was, as a compile-time variant: same-box A/B on fp64, far-axis stand-alone passes 2-7 % faster
// ((1024,256,512) axis 0 0.98 -> 0.89-0.95 ms), the near-stride pass of 1024^3 8 % slower (a wave's four
// 256-byte rows then lie 16 rows apart), fp32 and the fused pair slower; as a per-launch choice inside one
// kernel it cost 16 VGPRs + 100 bytes of scratch.  Not kept.)
template <int N, int R, bool COLS> struct WaveLocal { static constexpr bool value = !COLS && (64 % (N / R)) == 0; };

template <typename real, int N, int R, bool SPLIT, bool WL, int PADSH, int Ns, int... RADS> struct Stage;

template <typename real, int N, int R, bool SPLIT, bool WL, int PADSH, int Ns> struct Stage<real, N, R, SPLIT, WL, PADSH, Ns> {
  static __device__ __forceinline__ void run(cx<real> *, int, void *, const cx<real> *) {}
};

template <typename real, int N, int R, bool SPLIT, bool WL, int PADSH, int Ns, int r, int... REST>
struct Stage<real, N, R, SPLIT, WL, PADSH, Ns, r, REST...> {
  static constexpr int NT = N / R;   // threads per column
  static constexpr int NB = R / r;   // butterflies per thread in this stage
  static __device__ __forceinline__ void run(cx<real> *v, int t, void *col,
                                             const cx<real> *__restrict__ tw) {
    static_assert(R % r == 0 && N % (Ns * r) == 0, "bad radix plan");
    if constexpr (Ns > 1) {
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        const int k = (t + i * NT) % Ns;
        twiddle<real, N, r, Ns, NB>(v + i, k, tw);
      }
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) dft<real, r, NB>(v + i);
    [[maybe_unused]] constexpr int PH = Ns == 1 ? 8 : (Ns <= 32 ? 10 : 12);
    GFFT_PHASE(PH)
    if constexpr (sizeof...(REST) > 0) {
      // scatter: butterfly j = t + i*NT, output m -> index j0 + m*Ns, j0 = (j/Ns)*Ns*r + j%Ns.
      // pad_slot(j0 + m*Ns) == pad_slot(j0) + woff(m) and pad_slot(t + q*NT) == pad_slot(t) + roff(q)
      // for power-of-two Ns, NT: one address register per butterfly, the rest are DS immediates.
      constexpr int PAD = PADSH;
      int wbase[NB];
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        const int j = t + i * NT;
        wbase[i] = pad_slot<PAD>((j / Ns) * (Ns * r) + (j % Ns));
      }
      const int rbase = pad_slot<PAD>(t);
      if constexpr (SPLIT) {
        real *w = reinterpret_cast<real *>(col);
        tile_sync<WL>();
#pragma unroll
        for (int i = 0; i < NB; ++i)
#pragma unroll
          for (int m = 0; m < r; ++m) w[wbase[i] + pad_slot<PAD>(m * Ns)] = v[i + m * NB].x;
        tile_sync<WL>();
#pragma unroll
        for (int q = 0; q < R; ++q) v[q].x = w[rbase + pad_slot<PAD>(q * NT)];
        tile_sync<WL>();
#pragma unroll
        for (int i = 0; i < NB; ++i)
#pragma unroll
          for (int m = 0; m < r; ++m) w[wbase[i] + pad_slot<PAD>(m * Ns)] = v[i + m * NB].y;
        tile_sync<WL>();
#pragma unroll
        for (int q = 0; q < R; ++q) v[q].y = w[rbase + pad_slot<PAD>(q * NT)];
      } else {
        float2 *w = reinterpret_cast<float2 *>(col);
        tile_sync<WL>();
#pragma unroll
        for (int i = 0; i < NB; ++i)
#pragma unroll
          for (int m = 0; m < r; ++m)
            w[wbase[i] + pad_slot<PAD>(m * Ns)] = make_float2(v[i + m * NB].x, v[i + m * NB].y);
        tile_sync<WL>();
#pragma unroll
        for (int q = 0; q < R; ++q) {
          float2 f = w[rbase + pad_slot<PAD>(q * NT)];
          v[q].x = f.x;
          v[q].y = f.y;
        }
      }
      GFFT_PHASE(PH + 1)
      Stage<real, N, R, SPLIT, WL, PADSH, Ns * r, REST...>::run(v, t, col, tw);
    }
  }
};

// ---- stages that do not all keep the same number of values per thread ----------------------------------------
// A stage of radix r works on R_s = (R / r) * r values per thread: R itself when r divides R (every plan of rounds 1-4),
// fewer otherwise -- n = 960 = 15 x 16 x 4 on R = 16: the radix-15 stage keeps 15 values in each of 64 threads per
// column, the radix-16 and radix-4 stages 16 values in 60.  A column then takes TPC = max_s n / R_s threads, of which
// stage s uses the first n / R_s; the LDS exchange between two stages is written in the geometry of the one and read in
// the geometry of the other, and the pass loads in the first stage's geometry and stores in the last one's.
// (What makes 3 x 5 x 2^k lengths one-pass lengths: a single R would have to be a multiple of 15 AND of 16.)
template <int N, int R, int... RADS> struct Geo {
  static constexpr int cnt = (int)sizeof...(RADS);
  static constexpr int rad(int i) {
    constexpr int r[] = {RADS..., 1};
    return r[i];
  }
  static constexpr int Rs(int i) { return (R / rad(i)) * rad(i); }
  static constexpr int RL = cnt ? Rs(0) : R;                      // values per thread on the load side
  static constexpr int RS = cnt ? Rs(cnt ? cnt - 1 : 0) : R;      // ... on the store side
  static constexpr int tpc() {
    int m = N / R;
    for (int i = 0; i < cnt; ++i)
      if (N / Rs(i) > m) m = N / Rs(i);
    return m;
  }
  static constexpr int TPC = tpc();                               // threads per column
  static constexpr bool uniform() {
    for (int i = 0; i < cnt; ++i)
      if (Rs(i) != R) return false;
    return true;
  }
  static constexpr bool UNIFORM = uniform();
};

template <typename real, int N, int R, bool SPLIT, int Ns, int... RADS> struct StageV;
template <typename real, int N, int R, bool SPLIT, int Ns> struct StageV<real, N, R, SPLIT, Ns> {
  static __device__ __forceinline__ void run(cx<real> *, int, void *, const cx<real> *) {}
};
template <typename real, int N, int R, bool SPLIT, int Ns, int r, int... REST>
struct StageV<real, N, R, SPLIT, Ns, r, REST...> {
  static constexpr int RC = (R / r) * r;      // values per thread in this stage
  static constexpr int NT = N / RC;           // threads per column at work in it
  static constexpr int NB = RC / r;           // butterflies per thread
  static constexpr int RNX = FirstRadix<REST...>::value;
  static constexpr int RN = RNX ? (R / RNX) * RNX : RC;       // the next stage's values per thread ...
  static constexpr int NTN = N / RN;                          // ... and threads
  static __device__ __forceinline__ void run(cx<real> *v, int t, void *col, const cx<real> *__restrict__ tw) {
    static_assert(RC >= r && N % RC == 0 && N % (Ns * r) == 0, "bad radix plan");
    const bool mine = t < NT;
    if (mine) {
      if constexpr (Ns > 1) {
#pragma unroll
        for (int i = 0; i < NB; ++i) twiddle<real, N, r, Ns, NB>(v + i, (t + i * NT) % Ns, tw);
      }
#pragma unroll
      for (int i = 0; i < NB; ++i) dft<real, r, NB>(v + i);
    }
    if constexpr (sizeof...(REST) > 0) {
      // butterfly j = t + i NT, output m -> slot (j / Ns) Ns r + j % Ns + m Ns (the Stockham autosort scatter); the next
      // stage's thread t reads its slots t + q NTN.  No slot padding: these lengths scatter with odd multiples.
      int wbase[NB];
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        const int j = t + i * NT;
        wbase[i] = (j / Ns) * (Ns * r) + (j % Ns);
      }
      const bool next = t < NTN;
      if constexpr (SPLIT) {
        real *w = reinterpret_cast<real *>(col);
        __syncthreads();
        if (mine) {
#pragma unroll
          for (int i = 0; i < NB; ++i)
#pragma unroll
            for (int m = 0; m < r; ++m) w[wbase[i] + m * Ns] = v[i + m * NB].x;
        }
        __syncthreads();
        if (next) {
#pragma unroll
          for (int q = 0; q < RN; ++q) v[q].x = w[t + q * NTN];
        }
        __syncthreads();
        if (mine) {
#pragma unroll
          for (int i = 0; i < NB; ++i)
#pragma unroll
            for (int m = 0; m < r; ++m) w[wbase[i] + m * Ns] = v[i + m * NB].y;
        }
        __syncthreads();
        if (next) {
#pragma unroll
          for (int q = 0; q < RN; ++q) v[q].y = w[t + q * NTN];
        }
      } else {
        float2 *w = reinterpret_cast<float2 *>(col);
        __syncthreads();
        if (mine) {
#pragma unroll
          for (int i = 0; i < NB; ++i)
#pragma unroll
            for (int m = 0; m < r; ++m) w[wbase[i] + m * Ns] = make_float2(v[i + m * NB].x, v[i + m * NB].y);
        }
        __syncthreads();
        if (next) {
#pragma unroll
          for (int q = 0; q < RN; ++q) {
            const float2 f = w[t + q * NTN];
            v[q].x = f.x;
            v[q].y = f.y;
          }
        }
      }
      StageV<real, N, R, SPLIT, Ns * r, REST...>::run(v, t, col, tw);
    }
  }
};

// Tile-specialised load/store.  MODE is compile time (no per-element branches), BIGTW too.
// Conjugation (inverse transforms) is a sign on the imaginary part: on load one multiply, on
// store it is folded into the scale (scale_y = +-scale), so backward costs what forward does.
template <typename real> struct Vec2;
template <> struct Vec2<double> { typedef double type __attribute__((ext_vector_type(2))); };
template <> struct Vec2<float> { typedef float type __attribute__((ext_vector_type(2))); };

template <typename real, bool NT>
__device__ __forceinline__ cx<real> ldc(const cx<real> *p) {
  if constexpr (NT) {
    typename Vec2<real>::type t = __builtin_nontemporal_load(reinterpret_cast<const typename Vec2<real>::type *>(p));
    return {t.x, t.y};
  } else {
    return *p;
  }
}
template <typename real, bool NT>
__device__ __forceinline__ void stc(cx<real> *p, cx<real> v) {
  if constexpr (NT) {
    typename Vec2<real>::type t = {v.x, v.y};
    __builtin_nontemporal_store(t, reinterpret_cast<typename Vec2<real>::type *>(p));
  } else {
    *p = v;
  }
}

// `idx` = element index of (column, e) = base + e*in_es, built incrementally by the caller (one
// 64-bit add per element from a workgroup-uniform step: nothing per-element is loop invariant,
// so nothing gets hoisted out of the tile loop into long-lived VGPRs -- that hoisting cost
// 2 VGPRs per element per side, 64 VGPRs at R = 16).
template <typename real, int MODE, bool NT>
__device__ __forceinline__ cx<real> tile_load(const PassDesc &d, const void *__restrict__ in,
                                              int64_t base, int64_t idx, int e, real sy) {
  cx<real> v;
  if constexpr (MODE == MODE_R2R) {
    // real entry j = e - pos0 times its pre-factor; zero outside the line (unconditional load of
    // the line's first entry there, zeroed by a select)
    const int j = e - d.r2r_pos0;
    const bool ok = (unsigned)j < (unsigned)d.r2r_n;
    const real x = reinterpret_cast<const real *>(in)[ok ? idx - (int64_t)d.r2r_pos0 * d.in_es : base];
    const cx<real> a = reinterpret_cast<const cx<real> *>(d.r2r_pre)[ok ? j : 0];
    v = {ok ? a.x * x : (real)0, ok ? a.y * x : (real)0};
  } else if constexpr (MODE == MODE_R2C) {
    v.x = reinterpret_cast<const real *>(in)[idx];
    v.y = 0;
  } else if constexpr (MODE == MODE_C2R) {
    const int h = d.n >> 1;
    const bool mirror = e > h;
    const int ee = mirror ? d.n - e : e;
    v = ldc<real, NT>(reinterpret_cast<const cx<real> *>(in) + base + (int64_t)ee * d.in_es);
    v.y *= mirror ? -sy : sy;
  } else {
    v = ldc<real, NT>(reinterpret_cast<const cx<real> *>(in) + idx);
    v.y *= sy;
  }
  return v;
}

template <typename real, int MODE, bool BIGTW, bool NT>
__device__ __forceinline__ void tile_store(const PassDesc &d, void *__restrict__ out, int64_t idx,
                                           int e, unsigned mid, cx<real> v, real sx, real sy) {
  if constexpr (BIGTW) {
    const unsigned x = mid * (unsigned)e;          // < big_n <= 2^24
    const cx<real> a = reinterpret_cast<const cx<real> *>(d.tw_hi)[x >> d.tw_L];
    const cx<real> b = reinterpret_cast<const cx<real> *>(d.tw_lo)[x & ((1u << d.tw_L) - 1)];
    v = cmul(v, cmul(a, b));
  }
  if constexpr (MODE == MODE_R2R) {
    const int k = e - d.r2r_idx0;
    if ((unsigned)k < (unsigned)d.r2r_n) {
      const cx<real> b = reinterpret_cast<const cx<real> *>(d.r2r_post)[k];
      reinterpret_cast<real *>(out)[idx - (int64_t)d.r2r_idx0 * d.out_es] = (b.x * v.x - b.y * v.y) * sx;
    }
    return;
  }
  v.x *= sx;
  v.y *= sy;
  if constexpr (MODE == MODE_C2R) {
    reinterpret_cast<real *>(out)[idx] = v.x;
  } else if constexpr (MODE == MODE_R2C) {
    if (e <= (d.n >> 1)) stc<real, NT>(reinterpret_cast<cx<real> *>(out) + idx, v);
  } else {
    stc<real, NT>(reinterpret_cast<cx<real> *>(out) + idx, v);
  }
}

// ---- fused 3/2-rule truncation (store side, forward) and zero-padding (load side, backward) ----
// libfft.py:263-311 as load/store adapters: the padded transform of length n reads from / writes
// to the TRUNCATED array directly (d.tr_n entries along the axis), so the separate strided copies
// of the reference (and the full-size padded spectrum) disappear.  d.tr_N = truncated logical
// length (complex axis; for the real half-axis tr_N = tr_n), d.tr_even = the reference's parity
// rule (`N0 % 2 == 0`) applies.
template <typename real, int MODE>
__device__ __forceinline__ cx<real> tile_load_pad(const PassDesc &d, const void *__restrict__ in,
                                                  int64_t base, int64_t idx, int64_t shift, int e, real sy) {
  // idx = base + e*in_es (built incrementally by the caller); the upper half of the padded
  // spectrum sits `shift` = (n - tr_N)*in_es elements lower in the truncated array
  // Loads are unconditional (entries outside the kept band read the line's first element and are
  // zeroed by a select): a load inside a branch cannot be issued ahead of the previous one, and a
  // pass that waits for R loads one after the other is latency bound (measured on the backward
  // row pass of a padded 1024^3: 8.2 ms branching).
  // ... and everything after the load is arithmetic on select-computed factors: a conditional
  // `v *= 0.5` behind the load makes the compiler wait for THAT load inside the conditional, i.e.
  // before it issues the next one (the backward strided pass of a padded 1024^3 took 7.4 ms that
  // way against 6.7 ms unpadded, with a third of the data less to read).
  cx<real> v;
  if constexpr (MODE == MODE_C2R) {
    const cx<real> *p = reinterpret_cast<const cx<real> *>(in) + base;
    const bool mirror = e > (d.n >> 1);
    const int ee = mirror ? d.n - e : e;
    const bool ok = ee < d.tr_n;
    const bool nyq = d.tr_even && ee == d.tr_n - 1;
    v = p[ok ? (int64_t)ee * d.in_es : 0];
    const real fx = ok ? (nyq ? (real)0.5 : (real)1) : (real)0;
    const real fy = (ok && !nyq) ? (mirror ? -sy : sy) : (real)0;
    v.x *= fx;
    v.y *= fy;
  } else {
    const int h = d.tr_N >> 1;
    const bool lo = e <= h, hi = h > 0 && e >= d.n - h;
    const bool half = d.tr_even && (e == h || e == d.n - h);
    // kept entries stored as equal blocks of an all-to-all buffer (PassDesc::tr_jump): a 32-bit block offset
    // (gfft_plan_set_split / gfft_plan_create_guru_padded keep it below 2^31 elements) -- computed in 64 bits
    // behind a test of tr_jump it cost the zero-padding kernels 16 VGPRs
    const int blk = ((lo ? e : e - (d.n - d.tr_N)) >> d.tr_lgper) * (int)d.tr_jump;
    const int64_t at = (lo ? idx : (hi ? idx - shift : base)) + ((lo || hi) ? blk : 0);
    v = reinterpret_cast<const cx<real> *>(in)[at];
    const real f = (lo || hi) ? (half ? (real)0.5 : (real)1) : (real)0;
    v.x *= f;
    v.y *= f * sy;
  }
  return v;
}

template <typename real, int MODE>
__device__ __forceinline__ void tile_store_trunc(const PassDesc &d, void *__restrict__ out, int64_t idx,
                                                 int64_t shift, int e, cx<real> v, real sx, real sy) {
  // idx = out0 + e*out_es (incremental); shift = (n - tr_N)*out_es for the upper half
  cx<real> *p = reinterpret_cast<cx<real> *>(out);
  if constexpr (MODE == MODE_R2C) {
    if (e < d.tr_n) {
      if (d.tr_even && e == d.tr_n - 1) { v.x *= 2; v.y = 0; }
      p[idx] = {v.x * sx, v.y * sy};
    }
  } else {
    const int h = d.tr_N >> 1;
    const bool lo = e <= h, hi = h > 0 && e >= d.n - h;
    if (d.tr_even && e == d.n - h) return;            // folded onto entry h by the kernel body
    if (lo || hi) {
      const int blk = ((lo ? e : e - (d.n - d.tr_N)) >> d.tr_lgper) * (int)d.tr_jump;
      p[(lo ? idx : idx - shift) + blk] = {v.x * sx, v.y * sy};
    }
  }
}

// ---- real transforms of length 2N as one complex transform of length N (MODE_R2C_H / MODE_C2R_H) ----
// The real line is read / written as the complex line z[j] = x[2j] + i x[2j+1] (16-byte loads in
// fp64, half the butterflies of the full-length-with-zero-imaginary form), and the spectrum of x
// follows from Z = DFT_N z by the Hermitian pass
//     X[k] = (Z[k] + conj Z[N-k]) / 2  -  (i/2) w^k (Z[k] - conj Z[N-k]),   w = exp(-2 pi i / 2N),
// k = 0..N (Z[N] = Z[0]); the backward direction inverts it,
//     Z[k] = (X[k] + conj X[N-k])  +  i conj(w^k) (X[k] - conj X[N-k]).
// Both need entry N-k next to entry k.  A thread holds entries e = t + q NT; the mirror entries
// live in other threads, so the line passes through the LDS exchange buffer once more: every
// thread writes its R entries to their slots (thread 0 also fills slot N: Z[0] again, or X[N]),
// and reads slot N - e for each of them.  With split planes the real parts of the mirrors wait in
// R registers while the imaginary plane goes through.
// exp(-i pi q / R) for q < R as compile-time constants: the Hermitian twiddle of entry e = t + q NT
// is w^e = w^t exp(-i pi q / R) (NT = N / R, w = exp(-2 pi i / 2N)), so a thread loads ONE table entry
// per line and derives the rest by a multiplication with a literal -- R table loads per line cost R
// address / value register pairs that the scheduler hoists to the front.  The literals come from a
// constexpr Taylor evaluation in long double (R = 4 ... 24: powers of two, 12, 20, 24).
namespace ct {
constexpr long double PI = 3.14159265358979323846264338327950288L;
constexpr long double sin_small(long double x) {      // |x| <= pi/4
  long double term = x, sum = x;
  for (int k = 1; k < 16; ++k) { term *= -x * x / ((2 * k) * (2 * k + 1)); sum += term; }
  return sum;
}
constexpr long double cos_small(long double x) {
  long double term = 1, sum = 1;
  for (int k = 1; k < 16; ++k) { term *= -x * x / ((2 * k - 1) * (2 * k)); sum += term; }
  return sum;
}
constexpr long double cos_0pi(long double x) {        // 0 <= x <= pi
  return x <= PI / 4 ? cos_small(x) : (x <= 3 * PI / 4 ? -sin_small(x - PI / 2) : -cos_small(PI - x));
}
constexpr long double sin_0pi(long double x) {
  return x <= PI / 4 ? sin_small(x) : (x <= 3 * PI / 4 ? cos_small(x - PI / 2) : sin_small(PI - x));
}
}  // namespace ct
template <typename real, int R> struct MirrorTab {
  real c[R], s[R];
  constexpr MirrorTab() : c(), s() {
    for (int q = 0; q < R; ++q) {
      c[q] = (real)ct::cos_0pi(ct::PI * q / R);
      s[q] = (real)-ct::sin_0pi(ct::PI * q / R);
    }
  }
};
template <typename real, int R> __device__ __forceinline__ cx<real> mirror_twiddle(cx<real> wt, int q) {
  constexpr MirrorTab<real, R> tab{};
  const real c = tab.c[q], s = tab.s[q];
  return {wt.x * c - wt.y * s, wt.x * s + wt.y * c};
}

// (`active`: false for the thread rows of a column that hold no values in this geometry -- plans whose stages keep different
// numbers of values per thread, Geo -- : they take part in the barriers only)
template <typename real, int N, int R, bool SPLIT, bool WL, int PADSH, typename F>
__device__ __forceinline__ void mirror_pass(cx<real> *v, int t, void *col, cx<real> top, F &&combine, bool active = true) {
  constexpr int NT = N / R;
  constexpr int PAD = PADSH;
  if constexpr (SPLIT) {
    real *w = reinterpret_cast<real *>(col);
    real px[R];
    tile_sync<WL>();
    if (active) {
#pragma unroll
      for (int q = 0; q < R; ++q) w[pad_slot<PAD>(t + q * NT)] = v[q].x;
      if (t == 0) w[pad_slot<PAD>(N)] = top.x;
    }
    tile_sync<WL>();
    if (active) {
#pragma unroll
      for (int q = 0; q < R; ++q) px[q] = w[pad_slot<PAD>(N - (t + q * NT))];
    }
    tile_sync<WL>();
    if (active) {
#pragma unroll
      for (int q = 0; q < R; ++q) w[pad_slot<PAD>(t + q * NT)] = v[q].y;
      if (t == 0) w[pad_slot<PAD>(N)] = top.y;
    }
    tile_sync<WL>();
    if (active) {
#pragma unroll
      for (int q = 0; q < R; ++q) {
        const real py = w[pad_slot<PAD>(N - (t + q * NT))];
        combine(q, cx<real>{px[q], py});
      }
    }
  } else {
    float2 *w = reinterpret_cast<float2 *>(col);
    tile_sync<WL>();
    if (active) {
#pragma unroll
      for (int q = 0; q < R; ++q) w[pad_slot<PAD>(t + q * NT)] = make_float2(v[q].x, v[q].y);
      if (t == 0) w[pad_slot<PAD>(N)] = make_float2(top.x, top.y);
    }
    tile_sync<WL>();
    if (active) {
#pragma unroll
      for (int q = 0; q < R; ++q) {
        const float2 p = w[pad_slot<PAD>(N - (t + q * NT))];
        combine(q, cx<real>{(real)p.x, (real)p.y});
      }
    }
  }
}

// Uneven all-to-all blocks on the half-spectrum side of a packed-real row pass (FLAGS & 128, see
// PassDesc::ub_*): element offset of entry e of row `row`.  The block of e comes from a chain of at
// most seven compares against scalar boundaries; no divisions, no table look-ups.
__device__ __forceinline__ int64_t uneven_offset(const PassDesc &d, unsigned row, unsigned slab, unsigned srow, int e) {
  int start = 0, width = d.ub_start[1];
  int64_t base = d.ub_base[0];
#pragma unroll
  for (int b = 1; b < 8; ++b) {
    if (b < d.ub_p && e >= d.ub_start[b]) {
      start = d.ub_start[b];
      width = d.ub_start[b + 1] - d.ub_start[b];
      base = d.ub_base[b];
    }
  }
  const int ee = e - start;
  if (d.ub_n1 > 0) {
    // slab-wise blocks: rows of the body columns, then rows of the leftover columns (PassDesc::ub_n1);
    // 32-bit arithmetic (gfft_plan_set_split_slabs checks the buffer's size)
    const int lg = d.ub_tlg, bw = (width >> lg) << lg, n1 = (int)d.ub_n1;
    const int s0 = (int)base + (int)slab * (n1 * width);
    return ee < bw ? s0 + (int)srow * bw + ee : s0 + n1 * bw + (int)srow * (width - bw) + (ee - bw);
  }
  return d.ub_rows * (int64_t)start + (int64_t)row * width + ee;
}

// The same for the entries e = tl + q * NT_ a thread holds (q a compile-time slot index, NT_ threads per row): the block of
// the GROUP's first entry q * NT_ comes from scalar compares -- both sides are uniform --, and since no block is narrower than
// a group (PassDesc::ub_minw >= NT_, checked by the caller) at most ONE boundary falls inside it: one per-lane compare and
// selects between the two blocks' uniform quantities replace the chain of seven compare-and-select steps per entry, which
// cost the packed-real rows of config C5 a quarter of their time (stage 0 of a rank: 1.74 ms natural, 2.21 ms into the
// exchange buffer; profiles/r05_stage_probe.txt).
template <int NT_>
__device__ __forceinline__ int64_t uneven_offset_group(const PassDesc &d, unsigned row, unsigned slab, unsigned srow, int tl, int q) {
  const int e0 = q * NT_;
  int b = 0;
#pragma unroll
  for (int bb = 1; bb < 8; ++bb)
    if (bb < d.ub_p && e0 >= d.ub_start[bb]) b = bb;
  const int s_lo = d.ub_start[b], s_hi = d.ub_start[b + 1], s_top = d.ub_start[b + 2 > 8 ? 8 : b + 2];
  const int e = e0 + tl;
  const bool up = e >= s_hi;                    // (the last block ends at N + 1 > e: never)
  const int start = up ? s_hi : s_lo;
  const int ee = e - start;
  if (d.ub_n1 > 0) {
    const int lg = d.ub_tlg, n1 = (int)d.ub_n1;
    const int w_lo = s_hi - s_lo, w_hi = s_top - s_hi;
    const int bw_lo = (w_lo >> lg) << lg, bw_hi = (w_hi >> lg) << lg;
    const int s0_lo = (int)d.ub_base[b] + (int)slab * (n1 * w_lo), s0_hi = (int)d.ub_base[b + 1 > 7 ? 7 : b + 1] + (int)slab * (n1 * w_hi);
    const int s0 = up ? s0_hi : s0_lo, bw = up ? bw_hi : bw_lo, lw = up ? w_hi - bw_hi : w_lo - bw_lo;
    return ee < bw ? s0 + (int)srow * bw + ee : s0 + n1 * bw + (int)srow * lw + (ee - bw);
  }
  const int width = up ? s_top - s_hi : s_hi - s_lo;
  return d.ub_rows * (int64_t)start + (int64_t)row * width + ee;
}

template <int T, bool COLS, bool BIGTW>
__device__ __host__ __forceinline__ unsigned pow2_ntiles(const PassDesc &d) {
  const unsigned batch = (unsigned)d.batch, inner = (unsigned)d.inner, mid = (unsigned)d.mid;
  const unsigned flat_cols = mid * inner;
  if (COLS && !BIGTW) return (d.flat ? batch / flat_cols : batch / inner) * (((d.flat ? flat_cols : inner) + T - 1) / T);
  return (batch + T - 1) / T;
}

// Complex values at SYSTEM scope through raw buffer accesses (cache policy sc0 | sc1): element offsets
// are 32-bit, i.e. the hand-off buffer a tile addresses stays below 4 GiB (one plane of a fused pair)
#ifndef GFFT_HANDOFF_AUX
#define GFFT_HANDOFF_AUX 17      // cache policy of the hand-off accesses: 17 = sc0 | sc1 (system scope), 16 = sc1 (agent scope)
#endif
struct SysBuf {
  __amdgpu_buffer_rsrc_t r;
  __device__ __forceinline__ explicit SysBuf(const void *base)
      : r(__builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, 0x7fffffff, 0x00020000)) {}
  typedef unsigned u4 __attribute__((ext_vector_type(4)));
  typedef unsigned u2 __attribute__((ext_vector_type(2)));
  template <typename real> __device__ __forceinline__ cx<real> ld(int64_t elem) const {
    if constexpr (sizeof(real) == 8) return __builtin_bit_cast(cx<real>, __builtin_amdgcn_raw_buffer_load_b128(r, (unsigned)elem * 16u, 0, GFFT_HANDOFF_AUX));
    else return __builtin_bit_cast(cx<real>, __builtin_amdgcn_raw_buffer_load_b64(r, (unsigned)elem * 8u, 0, GFFT_HANDOFF_AUX));
  }
  template <typename real> __device__ __forceinline__ void st(int64_t elem, cx<real> v) const {
    if constexpr (sizeof(real) == 8) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, v), r, (unsigned)elem * 16u, 0, GFFT_HANDOFF_AUX);
    else __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2, v), r, (unsigned)elem * 8u, 0, GFFT_HANDOFF_AUX);
  }
};

// Unequal-width stage kernels (fft_mixv_f64.hip) in which this compiler (hipcc of ROCm 7.2) waits for EVERY load of a tile before
// it issues the next when the sign of conjugation-on-load is applied to each value as it arrives -- chains of 5 ... 17 loads each
// followed by s_waitcnt vmcnt(0), found by scanning the ISA (hipcc -S --cuda-device-only; tools/scan_serial_loads.py) --
// under their register pressure (scalar spills into VGPR lanes).  They apply the sign once, behind the loads (fft_pow2_body.inc
// DEFER_SIGN): 750^3 c128 23.5 -> 19.7 ms per step, 1050^3 60.2 -> 52.3, 1260^3 97.3 -> 84.9.  The kernels that were NOT
// serialised lose 1-2.5 % under the same change (840^3, 960^3, (720,1200,480)) and keep the sign on the load
// (profiles/r05_ab_serial_loads.txt).
constexpr bool in_lengths(int n, const int *list, int count) {
  for (int i = 0; i < count; ++i)
    if (list[i] == n) return true;
  return false;
}
constexpr bool serial_loads_f64(int n, bool cols, bool trunc) {
  constexpr int rows_plain[] = {112, 140, 280, 450, 540, 750, 900, 980, 1050, 1260, 1400, 1792, 2100, 2240, 2800, 3000, 3584};
  constexpr int rows_trunc[] = {1050, 1260, 1680, 2100};
  constexpr int cols_plain[] = {168, 250, 280, 350, 450, 750, 900, 1050, 1260, 1800, 2160};
  constexpr int cols_trunc[] = {150, 180, 450, 540, 588, 750, 840, 900, 1008, 1050, 1260, 1500, 1800, 2160, 2250, 2700};
  return cols ? (trunc ? in_lengths(n, cols_trunc, sizeof cols_trunc / sizeof(int)) : in_lengths(n, cols_plain, sizeof cols_plain / sizeof(int)))
              : (trunc ? in_lengths(n, rows_trunc, sizeof rows_trunc / sizeof(int)) : in_lengths(n, rows_plain, sizeof rows_plain / sizeof(int)));
}

// FLAGS: 1 = non-temporal loads, 2 = non-temporal stores, 4 = skip the transform (access-pattern
// probe), 8 = c2c only, 16 = fused truncation / padding adapters (d.tr_dir: 1 store, 2 load),
// 32 = transposing store (strided kernels whose OUTPUT is contiguous along the transform axis:
// the first pass of a four-step transform), 64 = with 16: the adapter is the zero-padding LOAD
// (backward direction) instead of the truncating STORE (a run-time direction switch inside the load
// loop serialises the loads), 128 = packed-real rows whose half-spectrum side is an all-to-all buffer
// of uneven blocks
// The pass over tiles  xcd_base + k,  k = k_first, k_first + k_step, ... < k_end  (the kernel below walks
// its share of all tiles; the fused two-pass kernel, fft_fused2_kernel, hands it one tile per ticket).
// FLAGS & 2048 / 4096: the output / input array is a hand-off buffer between workgroups of ONE launch
// (fused kernels): stored with / loaded at SYSTEM scope (sc0 sc1: written through to the memory side,
// never served from a possibly stale L2 line), plain complex accesses only; see fft_fused2_kernel.
struct NoHook { __device__ __forceinline__ void operator()() const {} };
// `after_loads`: called once per tile by every thread, after the tile's loads have been issued and before its
// first butterfly (the fused kernel settles the previous tile's hand-off there, behind the new loads)
template <typename real, int N, int R, int T, bool COLS, bool SPLIT, int FLAGS, int MODE, bool BIGTW, int... RADS, typename HOOK = NoHook>
__device__ __forceinline__ void pow2_body(const PassDesc &d, const void *__restrict__ in, void *__restrict__ out,
                                          unsigned char *smem, unsigned k_first, unsigned k_step, unsigned k_end,
                                          unsigned xcd_base, double scale, HOOK &&after_loads = HOOK()) {
#define GFFT_SCALE scale
#define GFFT_TILE_LOOP for (unsigned k = k_first; k < k_end; k += k_step)
#define GFFT_TILE_INDEX const unsigned tile = xcd_base + k;
#define GFFT_AFTER_LOADS after_loads();
#include "fft_pow2_body.inc"
#undef GFFT_TILE_LOOP
#undef GFFT_TILE_INDEX
#undef GFFT_AFTER_LOADS
#undef GFFT_SCALE
}

template <typename real, int N, int R, int T, bool COLS, bool SPLIT, int MINW, int FLAGS, int MODE, bool BIGTW, int... RADS>
__global__ void __launch_bounds__((T * Geo<N, R, RADS...>::TPC), MINW)
fft_pow2_kernel(PassDesc d, const void *__restrict__ in, void *__restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // Tile order.  Plain: tile = block + k*grid (adjacent tiles run at the same time on different
  // XCDs).  XCD-contiguous (d.swizzle, grid % 8 == 0): blocks are dealt to XCDs round-robin by
  // the dispatcher (block b -> XCD b % 8), so XCD x walks its own contiguous eighth of the tiles:
  // neighbouring column chunks then share one L2, which merges their partial cache lines when
  // rows are not line aligned (e.g. 513-wide half spectra).  Placement only affects speed.
  const unsigned per_xcd = (pow2_ntiles<T, COLS, BIGTW>(d) + 7) / 8;
  const unsigned kstep = d.swizzle ? gridDim.x / 8 : gridDim.x;
#define GFFT_TILE_LOOP for (unsigned k = d.swizzle ? blockIdx.x / 8 : blockIdx.x; k < (d.swizzle ? per_xcd : ntiles); k += kstep)
#define GFFT_TILE_INDEX const unsigned tile = d.swizzle ? (blockIdx.x % 8) * per_xcd + k : k;
#define GFFT_AFTER_LOADS
#define GFFT_SCALE d.scale
#include "fft_pow2_body.inc"
#undef GFFT_TILE_LOOP
#undef GFFT_TILE_INDEX
#undef GFFT_AFTER_LOADS
#undef GFFT_SCALE
}

// ---- two dependent passes in ONE persistent launch, handed over through the Infinity Cache -----------
// Two passes A -> B over an array far larger than the 256 MiB Infinity Cache normally cost 4 array-sized
// HBM transfers (A reads, A writes, B reads, B writes), and running them slab by slab as separate launches
// loses in launch tails what the cache gives back (DESIGN_HISTORY.md section 6).  Here both run inside one launch of
// one workgroup per CU.  The work is cut into PLANES -- a plane is a set of A tiles whose output is exactly
// the input of a set of B tiles (an i1-plane of a 3-D array for [rows along axis 2] -> [columns along axis
// 0]; one signal of a four-step transform) -- and A writes plane p into slot p % ring of a small ring buffer
// (ring x 16 MiB at 1024^2 complex128) that B reads back while it is still in the Infinity Cache: the
// intermediate array never travels to HBM and back.  Workgroups draw TICKETS from a global counter; the
// ticket order A(0) .. A(lag-1), B(0), A(lag), B(1), A(lag+1), ... makes every ticket wait only for lower
// tickets (B(p): all A tiles of plane p stored; A(p): all B tiles of plane p - ring done, the slot is free),
// so the launch cannot deadlock whatever the number of resident workgroups.
// Coherence between workgroups on different XCDs (whose L2s are not coherent within a launch) is per access:
// A stores the hand-off data at system scope (written through the L2), waits for the acknowledgements and
// only then raises the plane's counter; B spins on the counter with device-scope loads and reads the slot
// at system scope (FLAGS 2048 / 4096 of pow2_body).  Whole-L2 write-backs / invalidates (what an acquire /
// release pair compiles to) measured 2x SLOWER than two launches; this form 1.3-1.4x faster on a copy pair
// (tools/probes/mall_ring_probe.hip, profiles/r03_mall_ring_probe_*.txt).
// (struct FusedDesc: gfft_internal.h)

template <typename real_, int N, int R, int T, bool COLS, bool SPLIT, int FLAGS, int MODE, bool BIGTW, int... RADS>
struct PassCfg {
  typedef real_ real;
  static constexpr int threads = T * Geo<N, R, RADS...>::TPC;
  static constexpr int regs_per_thread = R * 2 * (int)(sizeof(real_) / 4);     // the column's values alone, in VGPRs
  static constexpr size_t lds = (size_t)T * Lds<N, COLS, T, (SPLIT && sizeof(real_) == 4), FirstRadix<RADS...>::value>::CS * (SPLIT ? sizeof(real_) : 2 * sizeof(real_));
  template <typename HOOK>
  static __device__ __forceinline__ void tile(const PassDesc &d, const void *in, void *out, unsigned char *smem, unsigned t,
                                              double scale, HOOK &&hook) {
    pow2_body<real_, N, R, T, COLS, SPLIT, FLAGS, MODE, BIGTW, RADS...>(d, in, out, smem, t, 1u, t + 1u, 0u, scale, hook);
  }
  static unsigned ntiles(const PassDesc &d) { return pow2_ntiles<T, COLS, BIGTW>(d); }
};

// (Everything one thread does on behalf of its workgroup -- drawing a ticket, polling a counter, raising
// one -- sits behind a WAVE-uniform scalar branch, `first lane of this wave is thread 0`, with the lane
// condition nested inside.  Written as a bare `if (threadIdx.x == 0)` in a loop whose body holds barriers,
// the compiler's control-flow structurizer is free to make lane 0 LEAVE the loop for its atomic while the
// other 63 lanes of its wave run on to the next barrier -- they then never see the new ticket: the first
// version of this kernel hung exactly so, with the barriers of the tile code in the loop.)
__device__ __forceinline__ bool wave_of_thread0() { return __builtin_amdgcn_readfirstlane(threadIdx.x) == 0; }

// A wait that takes longer than f.wait_ticks of the 100 MHz wall clock (a device shared with a long foreign kernel, a
// debugger single-stepping a workgroup) VOIDS the launch instead of hanging the device or poisoning the HIP context:
// the waiter raises the sticky word ctr[1], writes the plan's id into the host-visible word f.host_flag (pinned host
// memory: the library reads it at its next entry and re-plans the pair as two stand-alone passes, plan.cpp
// poll_async_error), and pushes the ticket counter past the last ticket so that every workgroup leaves at its next
// draw; the other waiters look at ctr[1] every 64 polls and leave too.  The output of such a launch is garbage and
// is reported as such -- GFFT_ERR_HIP from the library's next call -- but the process and its other plans live on.
// (wait_ticks == 0: every wait gives up at once -- the test hook for exactly this path.)
__device__ __forceinline__ void fused_give_up(const FusedDesc &f) {
  if (atomicAdd(&f.ctr[1], 1u) == 0u) {
    atomicAdd(&f.ctr[0], 0x80000000u);
    if (f.host_flag) {
      __hip_atomic_store(f.host_flag, f.plan_id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __threadfence_system();
    }
  }
}
__device__ __forceinline__ void fused_wait(const unsigned *flag, unsigned want, const FusedDesc &f) {
  if (wave_of_thread0()) {
    if (threadIdx.x == 0) {
      if (f.wait_ticks == 0u) {
        fused_give_up(f);
      } else {
        unsigned spins = 0;
        unsigned long long t0 = 0;
        while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
          __builtin_amdgcn_s_sleep(8);
          if ((++spins & 63u) == 0u) {
            if (__hip_atomic_load(&f.ctr[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;   // the launch is void already
            const unsigned long long now = wall_clock64();
            if (!t0) t0 = now;
            else if (now - t0 > (unsigned long long)f.wait_ticks) { fused_give_up(f); break; }
          }
        }
      }
    }
  }
  __syncthreads();
}

// GFFT_FUSE2_TRACE (a developer build, `make TRACE=1`): every workgroup stamps the 100 MHz wall clock at the
// phase boundaries of its first 96 tickets into a buffer behind the counters (tools/fused2_trace.py).
#ifdef GFFT_FUSE2_TRACE
#define GFFT_TRACE_SLOTS 96
#define GFFT_TRACE_STAMP(i)                                                                                    \
  if (it < GFFT_TRACE_SLOTS && wave_of_thread0()) {                                                              \
    if (threadIdx.x == 0) trace[((size_t)blockIdx.x * GFFT_TRACE_SLOTS + it) * 16 + (i)] = wall_clock64();      \
  }
#else
#define GFFT_TRACE_STAMP(i)
#endif

// workgroups of one fused launch resident per CU: two where both the LDS and the register file (128 VGPRs per thread
// then) allow it, else one -- which may use 256 VGPRs per thread when it has at most 512 threads
template <typename A, typename B> constexpr int fused2_per_cu() {
  constexpr size_t lds = A::lds > B::lds ? A::lds : B::lds;
  constexpr int budget = (A::regs_per_thread >= 64 || B::regs_per_thread >= 64) ? 256 : 128;     // VGPRs per thread the tiles want (64 of values alone: a radix-32 stage spills under 128)
  return (2 * A::threads / 256 * budget <= 512 && 2 * lds + 1024 <= 160 * 1024) ? 2 : 1;
}
// (workgroups of <= 512 threads: two per CU -- one computes while the other loads / stores --, which the
// register budget must allow: at most 128 VGPRs, i.e. 4 waves per SIMD)
template <typename A, typename B>
__global__ void __launch_bounds__(A::threads, (fused2_per_cu<A, B>() * A::threads / 256))
fft_fused2_kernel(const PassDesc *__restrict__ descs, FusedDesc f, double scale_a, double scale_b, const void *__restrict__ in,
                  void *__restrict__ ring, void *__restrict__ out) {
  static_assert(A::threads == B::threads, "both passes of a fused pair run on one workgroup shape");
  // (the two pass descriptors live in device memory: as by-value kernel arguments handed on by reference the
  // compiler materialised them whole, 200 spilled SGPRs; from memory every field is a scalar load at its use)
  const PassDesc &dA = descs[0], &dB = descs[1];
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ unsigned tk[4];          // [0], [1]: the ticket being worked on and the next one, drawn a tile ahead; [2]: a poll's result
  unsigned *done_a = f.ctr + 16, *done_b = f.ctr + 16 + f.planes;
  // (tickets count GROUPS of f.group consecutive tiles of one plane and pass)
  const unsigned grp = (unsigned)f.group;
  const unsigned ta = (unsigned)f.tiles_a / grp, tb = (unsigned)f.tiles_b / grp, per_pair = ta + tb;
  const unsigned head = (unsigned)f.lag * ta, pairs = (unsigned)(f.planes - f.lag);
  const unsigned total = (unsigned)f.planes * per_pair;
#ifdef GFFT_FUSE2_TRACE
  unsigned long long *trace = reinterpret_cast<unsigned long long *>(f.ctr + ((16 + 2 * (size_t)f.planes + 63) & ~(size_t)63));
  unsigned it = 0;
#endif
  // The next ticket is drawn while the current tile is being worked on (the atomic's round trip, ~2 us,
  // would otherwise sit between two tiles with the CU idle).  A workgroup then holds two tickets, the
  // lower one in work: the lowest unfinished ticket of the launch is still always in work somewhere, so
  // the no-deadlock argument stands.
  if (wave_of_thread0()) {
    if (threadIdx.x == 0) tk[0] = atomicAdd(&f.ctr[0], 1u);
  }
  // An A tile's counter may only go up once its write-through stores are acknowledged.  Waiting for that at
  // the end of the tile leaves the CU idle for a memory round trip; instead the tile leaves its counter OWED
  // (`owed` = the plane) and the workgroup settles it from inside the NEXT tile, right behind that tile's
  // loads (one s_waitcnt covers both) -- or before it starts polling a counter itself, so that it can never
  // wait, directly or through others, for its own debt (f.defer = 0: settle at the end of the tile).
  int owed = -1;
  auto settle = [&]() {
    if (owed >= 0) {
      __builtin_amdgcn_s_waitcnt(0);         // every write-through store of this wave acknowledged ...
      __syncthreads();                       // ... of every wave of the tile
      if (wave_of_thread0()) {
        if (threadIdx.x == 0) __hip_atomic_fetch_add(&done_a[owed], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      owed = -1;
    }
  };
  // a counter that is already there costs one poll; one that is not: settle the debt first, then wait
  auto await = [&](const unsigned *flag, unsigned want) {
    if (owed >= 0) {
      if (wave_of_thread0()) {
        if (threadIdx.x == 0) tk[2] = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= want ? 1u : 0u;
      }
      __syncthreads();
      const bool there = __builtin_amdgcn_readfirstlane(tk[2]) != 0;
      // (f.defer == 2: the debt is settled HERE, behind the next ticket's pick-up and the first look at its counter
      // -- the acknowledgements had those two round trips to arrive -- and before the tile's own loads)
      if (there) { if (f.defer == 2) settle(); return; }
      settle();
    }
    fused_wait(flag, want, f);
  };
#ifdef GFFT_FUSE2_TRACE
  auto hook = [&]() {
    settle();
    __builtin_amdgcn_s_waitcnt(0);           // (trace builds: the stamp marks the loads' ARRIVAL)
    GFFT_TRACE_STAMP(2)
  };
  for (it = 0;; ++it) {
#else
  auto &hook = settle;
  for (unsigned it = 0;; ++it) {
#endif
#ifdef GFFT_FUSE2_TRACE
    if (wave_of_thread0()) {
      if (threadIdx.x == 0) gfft_trace_slot = it < GFFT_TRACE_SLOTS ? trace + ((size_t)blockIdx.x * GFFT_TRACE_SLOTS + it) * 16 : nullptr;
    }
#endif
    __syncthreads();
    const unsigned k = __builtin_amdgcn_readfirstlane(tk[it & 1]);
    if (k >= total) break;
    GFFT_TRACE_STAMP(0)
    if (wave_of_thread0()) {
      if (threadIdx.x == 0) tk[(it + 1) & 1] = atomicAdd(&f.ctr[0], 1u);
    }
    bool is_a;
    unsigned p, t;
    if (k < head) {
      is_a = true; p = k / ta; t = k - p * ta;
    } else {
      const unsigned k2 = k - head;
      if (k2 < pairs * per_pair) {
        const unsigned j = k2 / per_pair, r = k2 - j * per_pair;
        if (r < tb) { is_a = false; p = j; t = r; } else { is_a = true; p = j + (unsigned)f.lag; t = r - tb; }
      } else {
        const unsigned k3 = k2 - pairs * per_pair, j = k3 / tb;
        is_a = false; p = pairs + j; t = k3 - j * tb;
      }
    }
    char *slot = static_cast<char *>(ring) + (size_t)(p % (unsigned)f.ring) * f.slot_bytes;
    if (f.debug == 2) continue;
#ifdef GFFT_FUSE2_TRACE
    if (it < GFFT_TRACE_SLOTS && wave_of_thread0()) {
      if (threadIdx.x == 0) trace[((size_t)blockIdx.x * GFFT_TRACE_SLOTS + it) * 16 + 7] = ((unsigned long long)k << 1) | (is_a ? 1u : 0u);
    }
#endif
    if (is_a) {
      if (p >= (unsigned)f.ring) await(&done_b[p - f.ring], tb);
      else if (f.defer == 2) settle();
      GFFT_TRACE_STAMP(1)
      if (f.debug != 3 && f.debug != 5)
        for (unsigned g = 0; g < grp; ++g)
          A::tile(dA, static_cast<const char *>(in) + (size_t)p * f.a_in_plane, slot, smem, t * grp + g, scale_a, hook);
      GFFT_TRACE_STAMP(3)
      owed = (int)p;
      if (!f.defer) settle();
      GFFT_TRACE_STAMP(4)
    } else {
      await(&done_a[p], ta);
      GFFT_TRACE_STAMP(1)
      if (f.debug != 3 && f.debug != 4)
        for (unsigned g = 0; g < grp; ++g)
          B::tile(dB, slot, static_cast<char *>(out) + (size_t)p * f.b_out_plane, smem, t * grp + g, scale_b, hook);
      GFFT_TRACE_STAMP(3)
      __syncthreads();
      if (wave_of_thread0()) {
        if (threadIdx.x == 0) __hip_atomic_fetch_add(&done_b[p], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      GFFT_TRACE_STAMP(4)
    }
  }
  settle();
}

// (Tried and not kept: the same launch with every WAVE on its own clock -- wave 0 draws tickets and polls the
// plane counters and posts both in a ring of LDS slots, each wave counts its arrived loads / acknowledged
// stores into the slot and the last one raises the plane counter, no workgroup barrier outside the strided
// tiles' exchanges -- so that a wave that is done starts the next tile's loads while its neighbours compute.
// Correct, 124 VGPRs, and 6 % SLOWER per step (34.33 -> 36.48 ms, profiles/r03_ab_fuse2_perwave.txt): the
// strided tiles' first exchange barrier just absorbs the drift (8.8 -> 11.2 us), i.e. the CU is not waiting
// for latency that overlap could hide.  A first form with every wave polling and raising the plane counters
// itself took 136 ms per step: sixteen memory-side atomics per tile on one hot word.)

template <typename A, typename B>
hipError_t launch_fused2(const PassDesc &dA, const PassDesc &dB, const PassDesc *dev_descs, const FusedDesc &f, const void *in,
                         void *ring, void *out, hipStream_t s) {
  constexpr size_t lds = A::lds > B::lds ? A::lds : B::lds;
  static_assert(lds <= 160 * 1024, "LDS budget");
  if (f.lag < 1 || f.ring <= f.lag || f.planes < 1 || f.tiles_a != (int)A::ntiles(dA) || f.tiles_b != (int)B::ntiles(dB) ||
      f.group < 1 || f.tiles_a % f.group || f.tiles_b % f.group)
    return hipErrorInvalidValue;
  auto kern = fft_fused2_kernel<A, B>;
  // (per device: a process may drive several GPUs, and both the function attribute and the CU count belong to one)
  static bool attr_set[kMaxDevices] = {};
  static int cus_of[kMaxDevices] = {};
  const int dev = current_device();
  if (!attr_set[dev] && lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    attr_set[dev] = true;
  }
  if (!cus_of[dev]) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1) { (void)hipGetLastError(); n = 256; }
    cus_of[dev] = n;
  }
  const int cus = cus_of[dev];
  hipError_t e = hipMemsetAsync(f.ctr, 0, (size_t)(16 + 2 * f.planes) * sizeof(unsigned), s);
  if (e != hipSuccess) return e;
  // as many workgroups as fit the CUs at once (the exchange tile of a 1024-thread workgroup fills the LDS,
  // two 512-thread ones share it): persistent, tickets do the load balancing
  constexpr int per_cu = fused2_per_cu<A, B>();
  hipLaunchKernelGGL(kern, dim3(cus * per_cu), dim3(A::threads), lds, s, dev_descs, f, dA.scale, dB.scale, in, ring, out);
  return hipGetLastError();
}

// ---- launch helpers ------------------------------------------------------------------------

template <typename real, int N, int R, int T, bool COLS, bool SPLIT, int MINW, int FLAGS, int MODE, bool BIGTW, int... RADS>
hipError_t launch_pow2_one(const PassDesc &d, const void *in, void *out, hipStream_t s) {
  constexpr int NT = Geo<N, R, RADS...>::TPC;
  constexpr int threads = T * NT;
  // (plans whose stages keep different numbers of values per thread: plain natural-layout complex passes only)
  if (!Geo<N, R, RADS...>::UNIFORM && (d.in_lgp || d.out_lgp || d.in_tlg || d.out_tlg || d.tw_hi || d.tr_jump)) return hipErrorInvalidValue;
  static_assert(threads >= 64 && threads <= 1024, "workgroup size");
  constexpr size_t lds_x = (sizeof...(RADS) > 1 || (FLAGS & 32) || MODE == MODE_R2C_H || MODE == MODE_C2R_H) ? (size_t)T * Lds<N, COLS, T, (SPLIT && sizeof(real) == 4), FirstRadix<RADS...>::value>::CS * (SPLIT ? sizeof(real) : 2 * sizeof(real)) : 0;
  constexpr size_t lds_f = (FLAGS & 16) ? (size_t)T * 2 * sizeof(real) : 0;
  constexpr size_t lds = lds_x > lds_f ? lds_x : lds_f;
  static_assert(lds <= 160 * 1024, "LDS budget");
  // split layouts: whole thread slots per block, plain complex passes only
  if (d.in_lgp || d.out_lgp) {
    const int lg = d.in_lgp > d.out_lgp ? d.in_lgp : d.out_lgp;
    if (((R >> lg) << lg) != R || MODE != MODE_C2C || BIGTW) return hipErrorInvalidValue;
    // (next to a fused truncation / padding the blocks sit on the PLAIN side only; the truncated side's
    // are PassDesc::tr_jump)
    if ((FLAGS & 16) && ((FLAGS & 64) ? d.in_lgp : d.out_lgp)) return hipErrorInvalidValue;
  }
  // tile-major lines (ROWS): thread slots advance by NT entries = whole tiles
  if (!COLS && ((d.in_tlg && (NT & ((1 << d.in_tlg) - 1))) || (d.out_tlg && (NT & ((1 << d.out_tlg) - 1)))))
    return hipErrorInvalidValue;
  auto kern = fft_pow2_kernel<real, N, R, T, COLS, SPLIT, MINW, FLAGS, MODE, BIGTW, RADS...>;
  if constexpr (lds > 64 * 1024) {
    static bool attr_set[kMaxDevices] = {};       // (per device, see launch_fused2)
    const int dev = current_device();
    if (!attr_set[dev]) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return e;
      attr_set[dev] = true;
    }
  }
  const int64_t flat_cols = d.mid * d.inner;
  const int64_t ntiles = (COLS && !BIGTW) ? (d.flat ? (d.batch / flat_cols) * ((flat_cols + T - 1) / T)
                                                    : (d.batch / d.inner) * ((d.inner + T - 1) / T))
                                          : (d.batch + T - 1) / T;
  const int64_t cap = d.grid_cap > 0 ? d.grid_cap : pow2_grid_cap();
  int grid = (int)(ntiles < cap ? ntiles : cap);
  PassDesc dd = d;
  if (dd.swizzle) {
    if (grid >= 64) grid = grid / 8 * 8; else dd.swizzle = 0;
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), lds, s, dd, in, out);
  return hipGetLastError();
}

// packed-real row plans (fft_real_*.hip): plain, with the truncating store (forward, 3/2-rule) or
// with the zero-padding load (backward), by d.tr_dir
template <typename real, int MODE, int N, int R, int T, bool SPLIT, int... RADS>
hipError_t half_launch_all(const PassDesc &d, const void *in, void *out, hipStream_t s);
#ifdef GFFT_VARIANTS
constexpr bool kFusedPadMix5 = true;
#else
constexpr bool kFusedPadMix5 = false;     // make VARIANTS=1 builds them (plan.cpp fused_pad_ok mirrors this)
#endif
template <typename real, int MODE, int N, int R, int T, bool SPLIT, int... RADS>
hipError_t half_launch(const PassDesc &d, const void *in, void *out, hipStream_t s) {
  static_assert(MODE == MODE_R2C_H || MODE == MODE_C2R_H, "packed-real modes");
  if constexpr (R == 20 && !kFusedPadMix5) {
    // 5^c 2^k lengths: plain and uneven-block kernels only
    if (d.tr_dir) return hipErrorInvalidValue;
    if (d.ub_p > 1) return launch_pow2_one<real, N, R, T, false, SPLIT, 1, 128, MODE, false, RADS...>(d, in, out, s);
    return launch_pow2_one<real, N, R, T, false, SPLIT, 1, 0, MODE, false, RADS...>(d, in, out, s);
  } else {
    return half_launch_all<real, MODE, N, R, T, SPLIT, RADS...>(d, in, out, s);
  }
}
template <typename real, int MODE, int N, int R, int T, bool SPLIT, int... RADS>
hipError_t half_launch_all(const PassDesc &d, const void *in, void *out, hipStream_t s) {
  if (d.ub_p > 1 && d.tr_dir == 0)
    return launch_pow2_one<real, N, R, T, false, SPLIT, 1, 128, MODE, false, RADS...>(d, in, out, s);
  if (d.ub_p > 1) {
    // ... of the KEPT entries of a truncated half spectrum (3/2-rule transforms on several ranks)
    if constexpr (MODE == MODE_R2C_H) {
      if (d.tr_dir == 1) return launch_pow2_one<real, N, R, T, false, SPLIT, 1, 16 | 128, MODE, false, RADS...>(d, in, out, s);
    } else {
      if (d.tr_dir == 2) return launch_pow2_one<real, N, R, T, false, SPLIT, 1, 16 | 64 | 128, MODE, false, RADS...>(d, in, out, s);
    }
    return hipErrorInvalidValue;
  }
  if (d.tr_dir == 0) return launch_pow2_one<real, N, R, T, false, SPLIT, 1, 0, MODE, false, RADS...>(d, in, out, s);
  if constexpr (MODE == MODE_R2C_H) {
    if (d.tr_dir == 1) return launch_pow2_one<real, N, R, T, false, SPLIT, 1, 16, MODE, false, RADS...>(d, in, out, s);
  } else {
    if (d.tr_dir == 2) return launch_pow2_one<real, N, R, T, false, SPLIT, 1, 16 | 64, MODE, false, RADS...>(d, in, out, s);
  }
  return hipErrorInvalidValue;
}

// runtime (mode, four-step twiddle) -> instantiation
// TABLE_FLAGS = kernel FLAGS, plus 512: this table entry is never picked for a four-step pass (the
// caller's condition excludes d.tw_hi), so its four-step-twiddle kernel is not instantiated; plus 1024: no
// fused truncation / zero-padding kernels for this entry (plan.cpp fused_pad_ok keeps such plans on the
// separate gfft_truncate / gfft_pad kernels)
template <typename real, int N, int R, int T, bool COLS, bool SPLIT, int MINW, int TABLE_FLAGS, int... RADS>
hipError_t launch_pow2_inst(const PassDesc &d, const void *in, void *out, hipStream_t s) {
  constexpr int FLAGS = TABLE_FLAGS & ~(512 | 1024);
  constexpr bool BIG_OK = COLS && !(TABLE_FLAGS & 512);     // (four-step passes always run strided, plan.cpp plan_fourstep)
  if constexpr ((TABLE_FLAGS & 1024) != 0) {
    if (d.tr_dir) return hipErrorInvalidValue;
  }
  // FLAGS & 8: complex-to-complex, no four-step twiddle (fewer instantiations of fat configurations)
  if constexpr ((FLAGS & 32) != 0) {
    // transposing store: complex strided pass, with or without the four-step twiddle
    if (d.mode != MODE_C2C || d.tr_dir || d.in_lgp || d.out_lgp || !d.tw_hi) return hipErrorInvalidValue;
    return launch_pow2_one<real, N, R, T, COLS, SPLIT, MINW, FLAGS, MODE_C2C, true, RADS...>(d, in, out, s);
  } else if constexpr (FLAGS != 0) {
    if (d.tw_hi) return hipErrorInvalidValue;
    if (d.mode != MODE_C2C || d.tr_dir) return hipErrorInvalidValue;
    return launch_pow2_one<real, N, R, T, COLS, SPLIT, MINW, FLAGS, MODE_C2C, false, RADS...>(d, in, out, s);
  } else {
    if (d.tw_hi) {
      if constexpr (!BIG_OK) return hipErrorInvalidValue;
      else {
        if (d.mode != MODE_C2C) return hipErrorInvalidValue;
        return launch_pow2_one<real, N, R, T, COLS, SPLIT, MINW, FLAGS, MODE_C2C, true, RADS...>(d, in, out, s);
      }
    }
    if constexpr (!(TABLE_FLAGS & 1024)) {
      if (d.tr_dir == 1 && d.mode == MODE_C2C) return launch_pow2_one<real, N, R, T, COLS, SPLIT, MINW, 16, MODE_C2C, false, RADS...>(d, in, out, s);
      if (d.tr_dir == 1 && d.mode == MODE_R2C) return launch_pow2_one<real, N, R, T, COLS, SPLIT, MINW, 16, MODE_R2C, false, RADS...>(d, in, out, s);
      if (d.tr_dir == 2 && d.mode == MODE_C2C) return launch_pow2_one<real, N, R, T, COLS, SPLIT, MINW, 16 | 64, MODE_C2C, false, RADS...>(d, in, out, s);
      if (d.tr_dir == 2 && d.mode == MODE_C2R) return launch_pow2_one<real, N, R, T, COLS, SPLIT, MINW, 16 | 64, MODE_C2R, false, RADS...>(d, in, out, s);
    }
    if (d.tr_dir) return hipErrorInvalidValue;
    switch (d.mode) {
      case MODE_C2C: return launch_pow2_one<real, N, R, T, COLS, SPLIT, MINW, FLAGS, MODE_C2C, false, RADS...>(d, in, out, s);
      case MODE_R2C: return launch_pow2_one<real, N, R, T, COLS, SPLIT, MINW, FLAGS, MODE_R2C, false, RADS...>(d, in, out, s);
      case MODE_C2R: return launch_pow2_one<real, N, R, T, COLS, SPLIT, MINW, FLAGS, MODE_C2R, false, RADS...>(d, in, out, s);
#ifdef GFFT_INST_R2R   // translation units whose tables also serve the real-to-real adapters
      case MODE_R2R: return launch_pow2_one<real, N, R, T, COLS, SPLIT, MINW, FLAGS, MODE_R2R, false, RADS...>(d, in, out, s);
#endif
    }
  }
  return hipErrorInvalidValue;
}

}  // namespace gfft
