// libgfft.so: C ABI (include/gfft.h) + the planner.
//
// The planner is the device-side counterpart of fftw_planxfftn() (mpi4py_fft/fftw/
// fftw_planxfftn.c:10-77): it turns (sizes, axes, kind) into one strided+batched 1-D pass per
// transformed axis -- {n, element stride, batch dims} exactly as the reference builds FFTW guru
// iodims (.c:25-47), but 64-bit -- and binds each pass to a kernel family:
//   * power-of-two n <= 4096        -> fft_pow2 kernels (ROWS if the axis is contiguous, else COLS)
//   * any other n <= generic limit  -> fft_generic (LDS mixed radix)
//   * larger composite n            -> four-step: two strided passes + fused twiddle
#include "../../include/gfft.h"
#include "gfft_internal.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

using namespace gfft;

namespace {

thread_local std::string g_last_error;

int fail(int code, const std::string &msg) {
  g_last_error = msg;
  return code;
}

int hip_fail(hipError_t e, const char *what) {
  return fail(e == hipErrorNoDevice || e == hipErrorInvalidDevice ? GFFT_ERR_NO_DEVICE : GFFT_ERR_HIP,
              std::string(what) + ": " + hipGetErrorString(e));
}

#define HIP_TRY(expr)                                  \
  do {                                                 \
    hipError_t _e = (expr);                            \
    if (_e != hipSuccess) return hip_fail(_e, #expr);  \
  } while (0)

bool g_device_checked = false;
int check_device() {
  if (g_device_checked) return GFFT_OK;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) {
    (void)hipGetLastError();
    return fail(GFFT_ERR_NO_DEVICE, "no HIP device available (libgfft has no host fallback)");
  }
  g_device_checked = true;
  return GFFT_OK;
}

// ---- tunables ---------------------------------------------------------------------------
struct Options {
  int grid_cap = 4096;
  int variant_rows = 0;
  int variant_cols = 0;
  int force_generic = 0;
  int xcd_swizzle = -1;      // XCD-contiguous tile order in the pow2 kernels: 0 off, 1 on, -1 auto
  int profile = 0;           // record HIP events around every pass (bench.py roofline leg)
  int fused3 = 1;            // reorder + padded-pitch workspace for 3-D all-axes plans
  int64_t fused3_min_bytes = 32 << 20;
  Options() {
    if (const char *s = getenv("GFFT_GRID_CAP")) grid_cap = atoi(s);
    if (const char *s = getenv("GFFT_VARIANT_ROWS")) variant_rows = atoi(s);
    if (const char *s = getenv("GFFT_VARIANT_COLS")) variant_cols = atoi(s);
    if (const char *s = getenv("GFFT_FORCE_GENERIC")) force_generic = atoi(s);
    if (const char *s = getenv("GFFT_FUSED3")) fused3 = atoi(s);
    if (const char *s = getenv("GFFT_XCD_SWIZZLE")) xcd_swizzle = atoi(s);
  }
};
Options &opts() {
  static Options o;
  return o;
}

// ---- twiddle tables (device resident, shared by plans) ------------------------------------
std::mutex g_tw_mutex;
std::map<std::pair<int64_t, int>, void *> g_tw_cache;       // (n, precision) -> W_n^k, k<n
struct BigTw { void *hi, *lo; int L; };
std::map<std::pair<int64_t, int>, BigTw> g_bigtw_cache;     // (big_n, precision)

// exp(-2 pi i k / n) for k in [k0, k0 + count*step) step `step`, in long double, stored as `precision`
int upload_twiddles(int64_t n, int64_t step, int64_t count, int precision, void **out) {
  const long double w = -2.0L * 3.14159265358979323846264338327950288L / (long double)n;
  std::vector<unsigned char> host((size_t)count * 2 * precision);
  for (int64_t j = 0; j < count; ++j) {
    const int64_t k = (j * step) % n;
    // reduce to the first octant for accuracy
    long double c, s;
    {
      const long double a = w * (long double)k;
      c = cosl(a);
      s = sinl(a);
      if (4 * k == n) { c = 0; s = -1; }
      else if (2 * k == n) { c = -1; s = 0; }
      else if (4 * k == 3 * n) { c = 0; s = 1; }
      else if (k == 0) { c = 1; s = 0; }
    }
    if (precision == 8) {
      double *p = reinterpret_cast<double *>(host.data()) + 2 * j;
      p[0] = (double)c;
      p[1] = (double)s;
    } else {
      float *p = reinterpret_cast<float *>(host.data()) + 2 * j;
      p[0] = (float)c;
      p[1] = (float)s;
    }
  }
  void *d = nullptr;
  HIP_TRY(hipMalloc(&d, host.size()));
  HIP_TRY(hipMemcpy(d, host.data(), host.size(), hipMemcpyHostToDevice));
  *out = d;
  return GFFT_OK;
}

int get_twiddles(int64_t n, int precision, const void **out) {
  std::lock_guard<std::mutex> lock(g_tw_mutex);
  auto key = std::make_pair(n, precision);
  auto it = g_tw_cache.find(key);
  if (it == g_tw_cache.end()) {
    void *d = nullptr;
    int rc = upload_twiddles(n, 1, n, precision, &d);
    if (rc) return rc;
    it = g_tw_cache.emplace(key, d).first;
  }
  *out = it->second;
  return GFFT_OK;
}

int get_bigtw(int64_t big_n, int precision, BigTw *out) {
  std::lock_guard<std::mutex> lock(g_tw_mutex);
  auto key = std::make_pair(big_n, precision);
  auto it = g_bigtw_cache.find(key);
  if (it == g_bigtw_cache.end()) {
    int L = 0;
    while (((int64_t)1 << (2 * L)) < big_n) ++L;     // 2^L ~ sqrt(big_n)
    const int64_t lo_n = (int64_t)1 << L;
    const int64_t hi_n = (big_n + lo_n - 1) / lo_n;
    BigTw t{nullptr, nullptr, L};
    int rc = upload_twiddles(big_n, 1, lo_n < big_n ? lo_n : big_n, precision, &t.lo);
    if (rc) return rc;
    rc = upload_twiddles(big_n, lo_n, hi_n, precision, &t.hi);
    if (rc) return rc;
    it = g_bigtw_cache.emplace(key, t).first;
  }
  *out = it->second;
  return GFFT_OK;
}

// ---- plan -------------------------------------------------------------------------------
enum Buf { BUF_IN = 0, BUF_OUT = 1, BUF_WS = 2 };

struct Pass {
  PassDesc d{};
  Factors f{};
  bool pow2 = false, cols = false;
  bool first_of_fourstep = false, second_of_fourstep = false;
  int src = BUF_IN, dst = BUF_OUT;   // where the logical axis pass reads / must leave its data
  bool real_scaled = false;          // this pass carries the plan's scale factor
};

bool is_pow2(int64_t n) { return n > 0 && (n & (n - 1)) == 0; }

// lengths served by the register-resident kernels: powers of two 16..4096 and 3^b*2^k 48..3456
bool pow2_ok(int64_t n, int precision) {
  if (opts().force_generic) return false;
  if (n > 4096) return false;
  if (mix3_supported((int)n)) return true;
  return precision == 8 ? pow2_supported_f64((int)n) : pow2_supported_f32((int)n);
}

bool factorize(int64_t n, Factors *f, int max_prime) {
  f->count = 0;
  auto push = [&](int r) { if (f->count < 24) f->r[f->count++] = r; };
  while (n % 4 == 0) { push(4); n /= 4; }
  while (n % 2 == 0) { push(2); n /= 2; }
  while (n % 3 == 0) { push(3); n /= 3; }
  for (int64_t p = 5; p * p <= n; p += 2)
    while (n % p == 0) { push((int)p); n /= p; }
  if (n > 1) push((int)n);
  for (int i = 0; i < f->count; ++i)
    if (f->r[i] > max_prime) return false;
  if (f->count == 0) { f->count = 1; f->r[0] = 1; }
  return true;
}

constexpr int GENERIC_MAX_PRIME = 1024;   // O(r^2) butterfly above this is not worth running

}  // namespace

struct gfft_plan_s {
  int ndims = 0, kind = 0, precision = 0;
  std::vector<int64_t> sizes_in, sizes_out;
  std::vector<int> axes;
  std::vector<Pass> passes;
  void *workspace = nullptr;
  size_t workspace_bytes = 0, need_workspace_bytes = 0;
  double flops = 0, bytes = 0;
  int variant_rows = 0, variant_cols = 0, xcd_swizzle = 0;
  bool fused3 = false;
  bool uses_ws = false;        // some pass reads/writes BUF_WS
  bool has_fourstep = false;
  size_t c2r_ws_bytes = 0, fourstep_off = 0;
  std::vector<std::vector<hipEvent_t>> prof;   // per execute: events before pass 0 and after each pass
};

namespace {

// Build the pass(es) for one transformed axis.  `shape_in/out`: array shapes seen by this pass.
int plan_axis(gfft_plan_s *pl, int axis, int mode, bool inverse, const std::vector<int64_t> &shape_in,
              const std::vector<int64_t> &shape_out, int src, int dst) {
  const int nd = pl->ndims;
  const int prec = pl->precision;
  const int64_t n = (mode == MODE_C2R) ? shape_out[axis] : shape_in[axis];
  int64_t outer = 1, inner = 1;
  for (int i = 0; i < axis; ++i) outer *= shape_in[i];
  for (int i = axis + 1; i < nd; ++i) inner *= shape_in[i];
  const int64_t nin = shape_in[axis], nout = shape_out[axis];
  const int64_t batch = outer * inner;
  if (batch >= ((int64_t)1 << 31) || n >= ((int64_t)1 << 31))
    return fail(GFFT_ERR_UNSUPPORTED, "batch or length exceeds 2^31");
  const double lines = (double)batch;
  pl->flops += (mode == MODE_C2C ? 1.0 : 0.5) * 5.0 * (double)n * std::log2((double)n > 1 ? (double)n : 2.0) * lines * (n > 1 ? 1 : 0);
  const double esz_in = (mode == MODE_R2C) ? prec : 2.0 * prec;
  const double esz_out = (mode == MODE_C2R) ? prec : 2.0 * prec;
  pl->bytes += lines * ((double)nin * esz_in + (double)nout * esz_out);

  Pass p;
  p.src = src;
  p.dst = dst;
  p.d.n = (int)n;
  p.d.mode = mode;
  p.d.conj_in = inverse ? 1 : 0;
  p.d.conj_out = (inverse && mode != MODE_C2R) ? 1 : 0;
  p.d.batch = batch;
  p.d.mid = 1;
  p.d.inner = inner;
  p.d.in_os = nin * inner;
  p.d.in_ms = 0;
  p.d.in_is = 1;
  p.d.in_es = inner;
  p.d.out_os = nout * inner;
  p.d.out_ms = 0;
  p.d.out_is = 1;
  p.d.out_es = inner;
  p.d.scale = 1.0;
  p.d.tw_hi = p.d.tw_lo = nullptr;
  p.d.big_n = 0;
  p.d.tw_L = 0;

  const int gmax = generic_max_n(prec);
  if (pow2_ok(n, prec)) {
    p.pow2 = true;
    p.cols = inner > 1;
    int rc = get_twiddles(n, prec, &p.d.tw);
    if (rc) return rc;
    pl->passes.push_back(p);
    return GFFT_OK;
  }
  if (n <= gmax && factorize(n, &p.f, GENERIC_MAX_PRIME)) {
    int rc = get_twiddles(n, prec, &p.d.tw);
    if (rc) return rc;
    pl->passes.push_back(p);
    return GFFT_OK;
  }
  // ---- four-step: n = n1 * n2
  if (mode != MODE_C2C)
    return fail(GFFT_ERR_UNSUPPORTED, "real transforms longer than the single-pass limit are not planned yet");
  if (n >= ((int64_t)1 << 24)) return fail(GFFT_ERR_UNSUPPORTED, "transform length >= 2^24");
  int64_t n1 = 0, n2 = 0;
  if (is_pow2(n)) {
    int lg = 0;
    while (((int64_t)1 << lg) < n) ++lg;
    n1 = (int64_t)1 << ((lg + 1) / 2);
    n2 = n / n1;
  } else {
    for (int64_t a = (int64_t)std::sqrt((double)n); a >= 2; --a)
      if (n % a == 0) {
        Factors fa, fb;
        if (n / a <= gmax && factorize(a, &fa, GENERIC_MAX_PRIME) && factorize(n / a, &fb, GENERIC_MAX_PRIME)) {
          n1 = n / a;
          n2 = a;
          break;
        }
      }
  }
  if (n1 == 0) return fail(GFFT_ERR_UNSUPPORTED, "length has a prime factor too large for this engine");
  BigTw bt;
  int rc = get_bigtw(n, prec, &bt);
  if (rc) return rc;
  // The intermediate lives in the plan workspace as tmp[o][i2][k1][i] with the i2-stride S2
  // padded off the power of two (same channel-aliasing argument as the 3-D workspace).
  const int64_t esz = 2 * prec;
  int64_t S2 = n1 * inner;
  if ((S2 * esz) % 2048 == 0) S2 += 256 / esz;
  // step 1: length-n1 transforms over i1 (stride n2*inner) for every (o, i2, i); output transposed
  Pass a = p;
  a.first_of_fourstep = true;
  a.d.n = (int)n1;
  a.d.batch = outer * n2 * inner;
  a.d.mid = n2;
  a.d.inner = inner;
  a.d.in_os = n * inner;
  a.d.in_ms = inner;
  a.d.in_is = 1;
  a.d.in_es = n2 * inner;
  a.d.out_os = n2 * S2;
  a.d.out_ms = S2;
  a.d.out_is = 1;
  a.d.out_es = inner;
  a.d.conj_in = inverse ? 1 : 0;
  a.d.conj_out = 0;
  a.d.tw_hi = bt.hi;
  a.d.tw_lo = bt.lo;
  a.d.tw_L = bt.L;
  a.d.big_n = n;
  // step 2: length-n2 transforms over i2 (stride S2) for every (o, k1, i); natural output
  Pass b = p;
  b.second_of_fourstep = true;
  b.d.n = (int)n2;
  b.d.batch = outer * n1 * inner;
  b.d.mid = 1;
  b.d.inner = n1 * inner;
  b.d.in_os = n2 * S2;
  b.d.in_is = 1;
  b.d.in_es = S2;
  b.d.out_os = n * inner;
  b.d.out_is = 1;
  b.d.out_es = n1 * inner;
  b.d.conj_in = 0;
  b.d.conj_out = inverse ? 1 : 0;
  for (Pass *q : {&a, &b}) {
    const int64_t m = q->d.n;
    if (pow2_ok(m, prec)) {
      q->pow2 = true;
      q->cols = true;
    } else if (!(m <= gmax && factorize(m, &q->f, GENERIC_MAX_PRIME))) {
      return fail(GFFT_ERR_UNSUPPORTED, "four-step factor not plannable");
    }
    rc = get_twiddles(m, prec, &q->d.tw);
    if (rc) return rc;
  }
  size_t bytes = (size_t)outer * n2 * S2 * esz;
  if (bytes > pl->need_workspace_bytes) pl->need_workspace_bytes = bytes;
  pl->has_fourstep = true;
  pl->passes.push_back(a);
  pl->passes.push_back(b);
  return GFFT_OK;
}


// ---- 3-D all-axes plans on one GPU: pass order + padded-pitch workspace --------------------
// Strided passes over power-of-two pitches alias onto few HBM channels (measured on MI355X,
// 1024^3 c128: the axis-0 pass takes 11.0 ms on natural strides, 7.4 ms when rows are pitched
// 256 B wider).  User-visible arrays must stay C-contiguous, so the plan routes the data through
// one internal workspace W whose rows carry that extra pitch, and orders the passes so that each
// user array is touched by the pass that tolerates its layout best:
//   forward / r2c :  axis2 (rows) IN -> W | axis0 (cols) W -> W in place | axis1 (cols) W -> OUT
//   backward / c2r:  axis1 (cols) IN -> W | axis0 (cols) W -> W in place | axis2 (rows) W -> OUT
// Each pass still reads and writes every element exactly once (algorithmic traffic only).
bool fused3_applicable(const gfft_plan_s *pl) {
  if (!opts().fused3 || pl->ndims != 3 || pl->axes.size() != 3) return false;
  const bool real = pl->kind == GFFT_R2C || pl->kind == GFFT_C2R;
  if (real && pl->axes.back() != 2) return false;
  const std::vector<int64_t> &full = (pl->kind == GFFT_C2R) ? pl->sizes_out : pl->sizes_in;
  for (int i = 0; i < 3; ++i)
    if (!pow2_ok(full[i], pl->precision)) return false;
  const int64_t bytes = full[0] * full[1] * full[2] * (real ? 1 : 2) * pl->precision;
  return bytes >= opts().fused3_min_bytes;
}

int plan_fused3(gfft_plan_s *pl) {
  const int prec = pl->precision;
  const bool real = pl->kind == GFFT_R2C || pl->kind == GFFT_C2R;
  const bool inverse = pl->kind == GFFT_C2C_BACKWARD || pl->kind == GFFT_C2R;
  const std::vector<int64_t> &full = (pl->kind == GFFT_C2R) ? pl->sizes_out : pl->sizes_in;
  const int64_t n0 = full[0], n1 = full[1], n2 = full[2];
  const int64_t nc = real ? n2 / 2 + 1 : n2;           // complex entries per row
  const int64_t esz = 2 * prec;
  // workspace row pitch: rows start on 128-B lines; +256 B when the pitch would be a multiple of 2 KiB
  const int64_t seg = 128 / esz;
  int64_t P = (nc + seg - 1) / seg * seg;
  if ((P * esz) % 2048 == 0) P += 256 / esz;
  pl->need_workspace_bytes = (size_t)(n0 * n1 * P * esz);

  auto base = [&](int n, int mode) {
    Pass p;
    p.pow2 = true;
    p.d.n = n;
    p.d.mode = mode;
    p.d.conj_in = inverse ? 1 : 0;
    p.d.conj_out = (inverse && mode != MODE_C2R) ? 1 : 0;
    p.d.scale = 1.0;
    p.d.mid = 1;
    p.d.inner = 1;
    p.d.in_ms = p.d.out_ms = 0;
    p.d.in_is = p.d.out_is = 1;
    return p;
  };
  // Workspace layout W[i1][i0][c] (row pitch P): axis 0 is the NEAR strided axis inside W
  // (stride P), axis 1 the far one (stride n0*P).  The in-place middle pass then runs on near
  // strides on both its sides, and the pass that touches the user's natural array does so along
  // axis 1, the near axis of the natural layout (measured: near pad->pad 7.2 ms, far 7.9 ms).
  const int64_t w_i0 = P, w_i1 = n0 * P;
  // rows: transform along axis 2; batch (o = i0, i = i1); strides in elements of each side's type
  auto rows = [&](int mode, bool in_ws, bool out_ws, int src, int dst) {
    Pass p = base((int)n2, mode);
    p.cols = false;
    p.d.batch = n0 * n1;
    p.d.inner = n1;
    const int64_t nat_in = (mode == MODE_R2C) ? n2 : nc, nat_out = (mode == MODE_C2R) ? n2 : nc;
    p.d.in_os = in_ws ? w_i0 : n1 * nat_in;   p.d.in_is = in_ws ? w_i1 : nat_in;   p.d.in_es = 1;
    p.d.out_os = out_ws ? w_i0 : n1 * nat_out; p.d.out_is = out_ws ? w_i1 : nat_out; p.d.out_es = 1;
    p.src = src; p.dst = dst;
    return p;
  };
  // axis 1: batch (o = i0, i = c)
  auto axis1 = [&](bool in_ws, bool out_ws, int src, int dst) {
    Pass p = base((int)n1, MODE_C2C);
    p.cols = true;
    p.d.batch = n0 * nc;
    p.d.inner = nc;
    p.d.in_os = in_ws ? w_i0 : n1 * nc;   p.d.in_es = in_ws ? w_i1 : nc;
    p.d.out_os = out_ws ? w_i0 : n1 * nc; p.d.out_es = out_ws ? w_i1 : nc;
    p.src = src; p.dst = dst;
    return p;
  };
  // axis 0 inside the workspace: batch (o = i1, i = c)
  auto axis0 = [&](int src, int dst) {
    Pass p = base((int)n0, MODE_C2C);
    p.cols = true;
    p.d.batch = n1 * nc;
    p.d.inner = nc;
    p.d.in_os = w_i1;  p.d.in_es = w_i0;
    p.d.out_os = w_i1; p.d.out_es = w_i0;
    p.src = src; p.dst = dst;
    return p;
  };
  std::vector<Pass> seq;
  if (!inverse) {
    seq.push_back(rows(real ? MODE_R2C : MODE_C2C, false, true, BUF_IN, BUF_WS));
    seq.push_back(axis0(BUF_WS, BUF_WS));
    seq.push_back(axis1(true, false, BUF_WS, BUF_OUT));
  } else {
    seq.push_back(axis1(false, true, BUF_IN, BUF_WS));
    seq.push_back(axis0(BUF_WS, BUF_WS));
    seq.push_back(rows(real ? MODE_C2R : MODE_C2C, true, false, BUF_WS, BUF_OUT));
  }
  for (Pass &p : seq) {
    int rc = get_twiddles(p.d.n, prec, &p.d.tw);
    if (rc) return rc;
    const double lines = (double)p.d.batch;
    const double n = p.d.n;
    pl->flops += (p.d.mode == MODE_C2C ? 1.0 : 0.5) * 5.0 * n * std::log2(n) * lines;
    const double ein = p.d.mode == MODE_R2C ? prec : esz, eout = p.d.mode == MODE_C2R ? prec : esz;
    const double nin = p.d.mode == MODE_C2R ? (double)nc : n, nout = p.d.mode == MODE_R2C ? (double)nc : n;
    pl->bytes += lines * (nin * ein + nout * eout);
    pl->passes.push_back(p);
  }
  pl->fused3 = true;
  return GFFT_OK;
}

hipError_t run_pass(const gfft_plan_s *pl, const Pass &p, const PassDesc &d0, const void *in, void *out, hipStream_t s) {
  PassDesc d = d0;
  // auto: only where a strided pass writes rows that do not start on 128-byte lines (odd-width
  // half spectra): neighbouring chunks then meet in one L2 and their partial lines merge
  // (measured 1024^3 r2c: 5.4 -> 5.0 ms fp64, 3.6 -> 2.6 ms fp32; neutral-to-slightly-negative on
  // aligned arrays, so it stays off there)
  const int64_t esz_out = (d.mode == MODE_C2R ? 1 : 2) * (int64_t)pl->precision;
  d.swizzle = pl->xcd_swizzle >= 0 ? pl->xcd_swizzle
                                   : (p.cols && ((d.out_es * esz_out) % 128 != 0) ? 1 : 0);
  if (p.pow2 && mix3_supported(d.n)) {
    return pl->precision == 8 ? launch_mix3_f64(d, p.cols, in, out, s) : launch_mix3_f32(d, p.cols, in, out, s);
  }
  if (p.pow2) {
    const int variant = p.cols ? pl->variant_cols : pl->variant_rows;
    return pl->precision == 8 ? launch_pow2_f64(d, p.cols, variant, in, out, s)
                              : launch_pow2_f32(d, p.cols, variant, in, out, s);
  }
  return launch_generic(d, p.f, pl->precision, in, out, s);
}

}  // namespace

namespace gfft {
int pow2_grid_cap() { return opts().grid_cap > 0 ? opts().grid_cap : 4096; }
}  // namespace gfft

extern "C" {

const char *gfft_strerror(int status) {
  switch (status) {
    case GFFT_OK: return "success";
    case GFFT_ERR_INVALID: return "invalid argument";
    case GFFT_ERR_UNSUPPORTED: return "unsupported transform";
    case GFFT_ERR_NO_DEVICE: return "no HIP device";
    case GFFT_ERR_HIP: return "HIP runtime error";
    case GFFT_ERR_NOMEM: return "out of memory";
  }
  return "unknown gfft status";
}

const char *gfft_last_error(void) { return g_last_error.c_str(); }

int gfft_version(void) { return 100; }

int gfft_device_count(int *count) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) {
    (void)hipGetLastError();
    if (count) *count = 0;
    return fail(GFFT_ERR_NO_DEVICE, "no HIP device available");
  }
  if (count) *count = n;
  return GFFT_OK;
}

int gfft_device_name(int device, char *buf, size_t len) {
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, device));
  snprintf(buf, len, "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
  return GFFT_OK;
}

int gfft_set_option(const char *key, int value) {
  if (!key) return fail(GFFT_ERR_INVALID, "null option name");
  if (!strcmp(key, "grid_cap")) opts().grid_cap = value;
  else if (!strcmp(key, "variant_rows")) opts().variant_rows = value;
  else if (!strcmp(key, "variant_cols")) opts().variant_cols = value;
  else if (!strcmp(key, "force_generic")) opts().force_generic = value;
  else if (!strcmp(key, "copy_nt")) gfft::g_copy_nt = value;
  else if (!strcmp(key, "fused3")) opts().fused3 = value;
  else if (!strcmp(key, "profile")) opts().profile = value;
  else if (!strcmp(key, "xcd_swizzle")) opts().xcd_swizzle = value;
  else if (!strcmp(key, "fused3_min_mib")) opts().fused3_min_bytes = (int64_t)value << 20;
  else return fail(GFFT_ERR_INVALID, std::string("unknown option ") + key);
  return GFFT_OK;
}

int gfft_plan_create(gfft_plan *plan, int ndims, const int64_t *sizes_in, const int64_t *sizes_out,
                     int naxes, const int *axes, int kind, int precision) {
  if (!plan || !sizes_in || !sizes_out || !axes) return fail(GFFT_ERR_INVALID, "null argument");
  *plan = nullptr;
  if (ndims < 1 || ndims > 16 || naxes < 1 || naxes > ndims) return fail(GFFT_ERR_INVALID, "bad ndims/naxes");
  if (precision != GFFT_F32 && precision != GFFT_F64) return fail(GFFT_ERR_INVALID, "precision must be 4 or 8");
  if (kind != GFFT_C2C_FORWARD && kind != GFFT_C2C_BACKWARD && kind != GFFT_R2C && kind != GFFT_C2R)
    return fail(GFFT_ERR_UNSUPPORTED, "only c2c / r2c / c2r kinds are implemented (r2r kinds are out of scope)");
  std::vector<int> ax(axes, axes + naxes);
  std::vector<char> seen(ndims, 0);
  for (int &a : ax) {
    if (a < 0) a += ndims;
    if (a < 0 || a >= ndims || seen[a]) return fail(GFFT_ERR_INVALID, "bad or repeated axis");
    seen[a] = 1;
  }
  const int last = ax.back();
  for (int i = 0; i < ndims; ++i) {
    if (sizes_in[i] < 1 || sizes_out[i] < 1) return fail(GFFT_ERR_INVALID, "sizes must be >= 1");
    if (i == last && kind == GFFT_R2C) {
      if (sizes_out[i] != sizes_in[i] / 2 + 1) return fail(GFFT_ERR_INVALID, "r2c: sizes_out[axis] must be n/2+1");
    } else if (i == last && kind == GFFT_C2R) {
      if (sizes_in[i] != sizes_out[i] / 2 + 1) return fail(GFFT_ERR_INVALID, "c2r: sizes_in[axis] must be n/2+1");
    } else if (sizes_in[i] != sizes_out[i]) {
      return fail(GFFT_ERR_INVALID, "sizes_in and sizes_out may differ only along the halved axis");
    }
  }
  int rc = check_device();
  if (rc) return rc;

  gfft_plan_s *pl = new gfft_plan_s;
  pl->ndims = ndims;
  pl->kind = kind;
  pl->precision = precision;
  pl->sizes_in.assign(sizes_in, sizes_in + ndims);
  pl->sizes_out.assign(sizes_out, sizes_out + ndims);
  pl->axes = ax;
  pl->variant_rows = opts().variant_rows;
  pl->variant_cols = opts().variant_cols;
  pl->xcd_swizzle = opts().xcd_swizzle;

  rc = GFFT_OK;
  if (fused3_applicable(pl)) {
    rc = plan_fused3(pl);
  } else if (kind == GFFT_C2C_FORWARD || kind == GFFT_C2C_BACKWARD) {
    const bool inv = kind == GFFT_C2C_BACKWARD;
    for (int i = naxes - 1; i >= 0 && !rc; --i)
      rc = plan_axis(pl, ax[i], MODE_C2C, inv, pl->sizes_in, pl->sizes_in, i == naxes - 1 ? BUF_IN : BUF_OUT, BUF_OUT);
  } else if (kind == GFFT_R2C) {
    rc = plan_axis(pl, last, MODE_R2C, false, pl->sizes_in, pl->sizes_out, BUF_IN, BUF_OUT);
    for (int i = naxes - 2; i >= 0 && !rc; --i)
      rc = plan_axis(pl, ax[i], MODE_C2C, false, pl->sizes_out, pl->sizes_out, BUF_OUT, BUF_OUT);
  } else {  // C2R: complex passes first (into / inside a workspace: the input is never written,
            // unlike FFTW's multi-dimensional c2r), then the real pass
    for (int i = 0; i <= naxes - 2 && !rc; ++i)
      rc = plan_axis(pl, ax[i], MODE_C2C, true, pl->sizes_in, pl->sizes_in, i == 0 ? BUF_IN : BUF_WS, BUF_WS);
    if (!rc) rc = plan_axis(pl, last, MODE_C2R, true, pl->sizes_in, pl->sizes_out, naxes > 1 ? BUF_WS : BUF_IN, BUF_OUT);
    if (naxes > 1) {
      size_t bytes = 2 * (size_t)precision;
      for (int i = 0; i < ndims; ++i) bytes *= (size_t)sizes_in[i];
      pl->uses_ws = true;
      pl->c2r_ws_bytes = bytes;
    }
  }
  if (rc) {
    delete pl;
    return rc;
  }
  if (pl->uses_ws && !pl->fused3) {
    // workspace layout: [complex passes of a multi-axis c2r][scratch of an in-place four-step axis]
    const size_t fs = pl->need_workspace_bytes;
    pl->fourstep_off = pl->has_fourstep ? pl->c2r_ws_bytes : 0;
    pl->need_workspace_bytes = pl->c2r_ws_bytes + fs;
  }
  // the scale factor rides on the last pass
  pl->passes.back().real_scaled = true;
  *plan = pl;
  return GFFT_OK;
}

int gfft_execute(gfft_plan pl, const void *d_in, void *d_out, double scale, void *stream) {
  if (!pl || !d_in || !d_out) return fail(GFFT_ERR_INVALID, "null argument");
  if ((pl->kind == GFFT_R2C || pl->kind == GFFT_C2R) && d_in == d_out)
    return fail(GFFT_ERR_INVALID, "in-place real transforms are not supported");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  void *bufs[3] = {const_cast<void *>(d_in), d_out, nullptr};
  void *ws4 = nullptr;   // second scratch for a four-step axis that runs inside the workspace
  if (pl->fused3 || pl->uses_ws) {
    if (pl->workspace_bytes < pl->need_workspace_bytes) {
      if (pl->workspace) HIP_TRY(hipFree(pl->workspace));
      pl->workspace = nullptr;
      pl->workspace_bytes = 0;
      hipError_t e = hipMalloc(&pl->workspace, pl->need_workspace_bytes);
      if (e == hipErrorOutOfMemory) return fail(GFFT_ERR_NOMEM, "workspace allocation failed");
      HIP_TRY(e);
      pl->workspace_bytes = pl->need_workspace_bytes;
    }
    bufs[2] = pl->workspace;
    if (pl->has_fourstep && pl->uses_ws && !pl->fused3) ws4 = static_cast<char *>(pl->workspace) + pl->fourstep_off;
  }
  std::vector<hipEvent_t> *ev = nullptr;
  if (opts().profile) {
    pl->prof.emplace_back();
    ev = &pl->prof.back();
    hipEvent_t e0;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventRecord(e0, s));
    ev->push_back(e0);
  }
  auto mark = [&]() -> hipError_t {
    if (!ev) return hipSuccess;
    hipEvent_t e;
    hipError_t rc = hipEventCreate(&e);
    if (rc != hipSuccess) return rc;
    rc = hipEventRecord(e, s);
    ev->push_back(e);
    return rc;
  };
  for (size_t i = 0; i < pl->passes.size(); ++i) {
    const Pass &p = pl->passes[i];
    PassDesc d = p.d;
    if (p.real_scaled) d.scale = scale;
    const void *src = bufs[p.src];
    void *dst = bufs[p.dst];
    if (p.first_of_fourstep) {
      // step 1 writes a different layout: never in place
      const Pass &p2 = pl->passes[i + 1];
      PassDesc d2 = p2.d;
      if (p2.real_scaled) d2.scale = scale;
      void *mid = ws4;
      if (!mid) {
        if (pl->workspace_bytes < pl->need_workspace_bytes) {
          if (pl->workspace) HIP_TRY(hipFree(pl->workspace));
          pl->workspace = nullptr;
          pl->workspace_bytes = 0;
          hipError_t e = hipMalloc(&pl->workspace, pl->need_workspace_bytes);
          if (e == hipErrorOutOfMemory) return fail(GFFT_ERR_NOMEM, "workspace allocation failed");
          HIP_TRY(e);
          pl->workspace_bytes = pl->need_workspace_bytes;
        }
        mid = pl->workspace;
      }
      HIP_TRY(run_pass(pl, p, d, src, mid, s));
      HIP_TRY(mark());
      HIP_TRY(run_pass(pl, p2, d2, mid, dst, s));
      HIP_TRY(mark());
      ++i;
      continue;
    }
    HIP_TRY(run_pass(pl, p, d, src, dst, s));
    HIP_TRY(mark());
  }
  return GFFT_OK;
}

/* Accumulated per-pass kernel time since the last call (needs option "profile" = 1 while
 * executing): ms[i] = total milliseconds of pass i, *executes = number of executes summed.
 * Synchronises on the recorded events and frees them. */
int gfft_plan_profile(gfft_plan pl, float *ms, int max_passes, int *executes) {
  if (!pl || !ms) return fail(GFFT_ERR_INVALID, "null argument");
  for (int i = 0; i < max_passes; ++i) ms[i] = 0.f;
  int n = 0;
  for (auto &ev : pl->prof) {
    if (ev.empty()) continue;
    HIP_TRY(hipEventSynchronize(ev.back()));
    for (size_t i = 0; i + 1 < ev.size(); ++i) {
      float t = 0.f;
      HIP_TRY(hipEventElapsedTime(&t, ev[i], ev[i + 1]));
      if ((int)i < max_passes) ms[i] += t;
    }
    for (hipEvent_t e : ev) (void)hipEventDestroy(e);
    ++n;
  }
  pl->prof.clear();
  if (executes) *executes = n;
  return GFFT_OK;
}

/* text name of the kernel family behind pass i ("pow2-rows", "pow2-cols", "generic") */
int gfft_plan_pass_info(gfft_plan pl, int i, char *buf, size_t len, double *bytes) {
  if (!pl || i < 0 || i >= (int)pl->passes.size()) return fail(GFFT_ERR_INVALID, "bad pass index");
  const Pass &p = pl->passes[i];
  snprintf(buf, len, "%s n=%d", p.pow2 ? (p.cols ? "pow2-cols" : "pow2-rows") : "generic", p.d.n);
  if (bytes) {
    const double esz = 2.0 * pl->precision;
    const double nc = p.d.mode == MODE_C2C ? p.d.n : p.d.n / 2 + 1;
    const double ein = p.d.mode == MODE_R2C ? p.d.n * (double)pl->precision : nc * esz;
    const double eout = p.d.mode == MODE_C2R ? p.d.n * (double)pl->precision : nc * esz;
    *bytes = (double)p.d.batch * (ein + eout);
  }
  return GFFT_OK;
}

int gfft_plan_destroy(gfft_plan pl) {
  if (!pl) return GFFT_OK;
  if (pl->workspace) (void)hipFree(pl->workspace);
  delete pl;
  return GFFT_OK;
}

int gfft_plan_describe(gfft_plan pl, char *buf, size_t len) {
  if (!pl || !buf || !len) return fail(GFFT_ERR_INVALID, "null argument");
  std::string s;
  char line[256];
  const char *kn = pl->kind == GFFT_C2C_FORWARD ? "c2c-forward" : pl->kind == GFFT_C2C_BACKWARD ? "c2c-backward"
                   : pl->kind == GFFT_R2C ? "r2c" : "c2r";
  snprintf(line, sizeof line, "gfft plan: %s %s, %d dims, %zu passes%s\n", kn, pl->precision == 8 ? "f64" : "f32",
           pl->ndims, pl->passes.size(), pl->fused3 ? " [3-D schedule: padded-pitch workspace]" : "");
  s += line;
  for (const Pass &p : pl->passes) {
    snprintf(line, sizeof line, "  n=%d batch=%lld (mid=%lld inner=%lld) es_in=%lld es_out=%lld kernel=%s%s%s\n", p.d.n,
             (long long)p.d.batch, (long long)p.d.mid, (long long)p.d.inner, (long long)p.d.in_es,
             (long long)p.d.out_es, p.pow2 ? (p.cols ? "pow2-cols" : "pow2-rows") : "generic",
             p.first_of_fourstep ? " [four-step 1/2, fused twiddle]" : p.second_of_fourstep ? " [four-step 2/2]" : "",
             p.real_scaled ? " [scale]" : "");
    s += line;
  }
  snprintf(buf, len, "%s", s.c_str());
  return GFFT_OK;
}

int gfft_plan_cost(gfft_plan pl, double *flops, double *bytes, int *launches) {
  if (!pl) return fail(GFFT_ERR_INVALID, "null plan");
  if (flops) *flops = pl->flops;
  if (bytes) *bytes = pl->bytes;
  if (launches) *launches = (int)pl->passes.size();
  return GFFT_OK;
}

static int collapse(int ndims, const int64_t *shape, int axis, int64_t *outer, int64_t *naxis, int64_t *inner) {
  if (!shape || ndims < 1 || axis < 0 || axis >= ndims) return fail(GFFT_ERR_INVALID, "bad shape/axis");
  *outer = *inner = 1;
  for (int i = 0; i < axis; ++i) *outer *= shape[i];
  for (int i = axis + 1; i < ndims; ++i) *inner *= shape[i];
  *naxis = shape[axis];
  return GFFT_OK;
}

int gfft_pack(const void *d_array, void *d_packed, int ndims, const int64_t *shape, int axis, int nparts,
              int itemsize, void *stream) {
  int64_t o, n, i;
  int rc = collapse(ndims, shape, axis, &o, &n, &i);
  if (rc) return rc;
  if (nparts < 1 || n < nparts) return fail(GFFT_ERR_INVALID, "axis shorter than the number of parts");
  if ((rc = check_device())) return rc;
  HIP_TRY(launch_pack(d_array, d_packed, o, n, i, nparts, itemsize, false, (hipStream_t)stream));
  return GFFT_OK;
}

int gfft_unpack(const void *d_packed, void *d_array, int ndims, const int64_t *shape, int axis, int nparts,
                int itemsize, void *stream) {
  int64_t o, n, i;
  int rc = collapse(ndims, shape, axis, &o, &n, &i);
  if (rc) return rc;
  if (nparts < 1 || n < nparts) return fail(GFFT_ERR_INVALID, "axis shorter than the number of parts");
  if ((rc = check_device())) return rc;
  HIP_TRY(launch_pack(d_packed, d_array, o, n, i, nparts, itemsize, true, (hipStream_t)stream));
  return GFFT_OK;
}

int gfft_truncate(const void *d_padded, void *d_trunc, int ndims, const int64_t *shape_padded, int axis,
                  int64_t n_trunc, int is_real, int precision, double scale, void *stream) {
  int64_t o, n, i;
  int rc = collapse(ndims, shape_padded, axis, &o, &n, &i);
  if (rc) return rc;
  if (n_trunc < 1 || n_trunc > n) return fail(GFFT_ERR_INVALID, "bad truncated length");
  if ((rc = check_device())) return rc;
  HIP_TRY(launch_trunc(d_padded, d_trunc, o, n, n_trunc, i, is_real, precision, scale, false, (hipStream_t)stream));
  return GFFT_OK;
}

int gfft_pad(const void *d_trunc, void *d_padded, int ndims, const int64_t *shape_padded, int axis,
             int64_t n_trunc, int is_real, int precision, void *stream) {
  int64_t o, n, i;
  int rc = collapse(ndims, shape_padded, axis, &o, &n, &i);
  if (rc) return rc;
  if (n_trunc < 1 || n_trunc > n) return fail(GFFT_ERR_INVALID, "bad truncated length");
  if ((rc = check_device())) return rc;
  HIP_TRY(launch_trunc(d_trunc, d_padded, o, n, n_trunc, i, is_real, precision, 1.0, true, (hipStream_t)stream));
  return GFFT_OK;
}

int gfft_scale(void *d_data, int64_t count, int precision, double scale, void *stream) {
  int rc = check_device();
  if (rc) return rc;
  HIP_TRY(launch_scale(d_data, count, precision, scale, (hipStream_t)stream));
  return GFFT_OK;
}

int gfft_malloc(void **d_ptr, size_t bytes) {
  int rc = check_device();
  if (rc) return rc;
  hipError_t e = hipMalloc(d_ptr, bytes);
  if (e == hipErrorOutOfMemory) return fail(GFFT_ERR_NOMEM, "hipMalloc: out of memory");
  HIP_TRY(e);
  return GFFT_OK;
}
int gfft_free(void *d_ptr) { HIP_TRY(hipFree(d_ptr)); return GFFT_OK; }
int gfft_memcpy_h2d(void *d, const void *h, size_t n, void *s) { HIP_TRY(hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, (hipStream_t)s)); return GFFT_OK; }
int gfft_memcpy_d2h(void *h, const void *d, size_t n, void *s) { HIP_TRY(hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, (hipStream_t)s)); return GFFT_OK; }
int gfft_memcpy_d2d(void *d, const void *s_, size_t n, void *s) { HIP_TRY(hipMemcpyAsync(d, s_, n, hipMemcpyDeviceToDevice, (hipStream_t)s)); return GFFT_OK; }
int gfft_stream_synchronize(void *s) { HIP_TRY(hipStreamSynchronize((hipStream_t)s)); return GFFT_OK; }

int gfft_event_create(void **event) {
  int rc = check_device();
  if (rc) return rc;
  hipEvent_t e;
  HIP_TRY(hipEventCreate(&e));
  *event = e;
  return GFFT_OK;
}
int gfft_event_record(void *event, void *stream) { HIP_TRY(hipEventRecord((hipEvent_t)event, (hipStream_t)stream)); return GFFT_OK; }
int gfft_event_elapsed_ms(void *start, void *stop, float *ms) {
  HIP_TRY(hipEventSynchronize((hipEvent_t)stop));
  HIP_TRY(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
  return GFFT_OK;
}
int gfft_event_destroy(void *event) { HIP_TRY(hipEventDestroy((hipEvent_t)event)); return GFFT_OK; }

/* developer probe: run ONE power-of-two pass with explicit batch geometry and strides
 * (element units), bypassing the planner.  geom = {n, outer, mid, inner, in_os, in_ms, in_is,
 * in_es, out_os, out_ms, out_is, out_es}.  Not part of the drop-in boundary. */
int gfft_debug_pass(const int64_t *geom, int precision, int cols, int variant, int inverse,
                    const void *d_in, void *d_out, void *stream) {
  int rc = check_device();
  if (rc) return rc;
  PassDesc d{};
  d.n = (int)geom[0];
  d.mode = MODE_C2C;
  d.conj_in = d.conj_out = inverse;
  d.mid = geom[2];
  d.inner = geom[3];
  d.batch = geom[1] * geom[2] * geom[3];
  d.in_os = geom[4]; d.in_ms = geom[5]; d.in_is = geom[6]; d.in_es = geom[7];
  d.out_os = geom[8]; d.out_ms = geom[9]; d.out_is = geom[10]; d.out_es = geom[11];
  d.scale = 1.0;
  rc = get_twiddles(d.n, precision, &d.tw);
  if (rc) return rc;
  hipError_t e = precision == 8 ? launch_pow2_f64(d, cols != 0, variant, d_in, d_out, (hipStream_t)stream)
                                : launch_pow2_f32(d, cols != 0, variant, d_in, d_out, (hipStream_t)stream);
  HIP_TRY(e);
  return GFFT_OK;
}

int gfft_probe_copy(const void *d_src, void *d_dst, size_t bytes, void *stream) {
  int rc = check_device();
  if (rc) return rc;
  HIP_TRY(launch_copy(d_src, d_dst, bytes, (hipStream_t)stream));
  return GFFT_OK;
}
int gfft_probe_tile_copy(const void *d_src, void *d_dst, int64_t outer, int64_t n, int64_t inner, int tcols, void *stream) {
  int rc = check_device();
  if (rc) return rc;
  HIP_TRY(launch_tile_copy(d_src, d_dst, outer, n, inner, tcols, (hipStream_t)stream));
  return GFFT_OK;
}

}  // extern "C"
