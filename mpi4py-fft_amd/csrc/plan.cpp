// libgfft.so: C ABI (include/gfft.h) + the planner.
//
// The planner is the device-side counterpart of fftw_planxfftn() (mpi4py_fft/fftw/
// fftw_planxfftn.c:10-77): it turns (sizes, axes, kind) into one strided+batched 1-D pass per
// transformed axis -- {n, element stride, batch dims} exactly as the reference builds FFTW guru
// iodims (.c:25-47), but 64-bit -- and binds each pass to a kernel family:
//   * n = 2^k (16..4096), 3^b 2^k (48..3456) -> register-resident kernels (fft_pow2_impl.h; ROWS if
//                                       the axis is contiguous, else COLS)
//   * other n <= 4096 with small primes  -> fft_generic (LDS mixed radix)
//   * longer composite n                 -> four-step: two strided passes + fused twiddle
//   * anything else (large prime factors, long real transforms) -> Bluestein / complex embedding
//   * 3-D all-axes plans on one rank     -> plan_fused3: pass order + padded-pitch workspace
#include "../../include/gfft.h"
#include "gfft_internal.h"

#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <tuple>
#include <mutex>
#include <string>
#include <thread>
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
  int grid_cap = 0;          // workgroups per launch at most; 0 = 4096, and 16384 for the passes of complex 3-D schedules
  int variant_rows = 0;
  int variant_cols = 0;
  int force_generic = 0;
  int xcd_swizzle = -1;      // XCD-contiguous tile order in the pow2 kernels: 0 off, 1 on, -1 auto
  int profile = 0;           // record HIP events around every pass (bench.py roofline leg)
  int fused3 = 1;            // reorder + padded-pitch workspace for 3-D all-axes plans
  int debug_flat = 0;        // gfft_debug_pass: tiles over the flattened (mid, inner) index
  int flat_out = 1;          // forward r2c 3-D plans: far-axis last pass with flattened tiles
  int mixv_variant = 0;      // A/B: alternative kernels of the unequal-width lengths (tools/gen_mixv_tables.py)
  int mixv = 1;              // one-pass kernels for 3 x 5 x 2^k lengths (fft_mixv_*.hip); 0: the two-pass plans of rounds 1-4 (A/B)
  int wtile = 1;             // tile-major workspace under the complex 3-D pair schedule (plan_fused3; A/B)
  int plane2d = 1;           // planes of 32^2 / 64^2 points: both passes in one launch, the plane in LDS (fft_plane2d.hip; 0: two passes, A/B)
  int fuse2 = 1;             // pass pairs in one persistent launch, handed over through the Infinity Cache (fft_fused_f64.hip)
  int fuse2_ring = 0, fuse2_lag = 0;   // slots of the hand-off ring / planes the producer runs ahead; 0 = auto (make_fused2)
  int fuse2_kinds = 510;     // which pairs (bit = FusedKind): measured per kind, see make_fused2
  int fuse2_f32 = 1;         // 1: complex64 pairs (fft_fused_f32.hip); 2: the real fp32 pairs too (fft_fused_real_f32.hip: measured level, off)
  int fuse2_wait_ms = 2000;  // wall-clock limit of one wait inside a fused launch before the launch is voided (0: at once -- test hook)
  int debug_tile_lg = 0, debug_tile_side = 0, debug_tile_stride = 0;   // gfft_debug_pass: tile-major lines (rows passes)
  int64_t fused3_min_bytes = 32 << 20;
  // TEST HOOK (tests/test_gpu_rounding_guard.py): twiddle tables uploaded while debug_tw_exp > 0 carry ONE wrong entry --
  // the real part of entry debug_tw_index (mod n) is off by 10^-debug_tw_exp -- under their own cache key, so that plans made
  // before / after are untouched.  What the rounding-level guards of the test suite must catch; never set by the product.
  int debug_tw_index = 1, debug_tw_exp = 0;
  Options() {
    if (const char *s = getenv("GFFT_GRID_CAP")) grid_cap = atoi(s);
    if (const char *s = getenv("GFFT_VARIANT_ROWS")) variant_rows = atoi(s);
    if (const char *s = getenv("GFFT_VARIANT_COLS")) variant_cols = atoi(s);
    if (const char *s = getenv("GFFT_FORCE_GENERIC")) force_generic = atoi(s);
    if (const char *s = getenv("GFFT_FUSED3")) fused3 = atoi(s);
    if (const char *s = getenv("GFFT_XCD_SWIZZLE")) xcd_swizzle = atoi(s);
    if (const char *s = getenv("GFFT_FUSE2")) fuse2 = atoi(s);
    if (const char *s = getenv("GFFT_FUSE2_RING")) fuse2_ring = atoi(s);
    if (const char *s = getenv("GFFT_FUSE2_LAG")) fuse2_lag = atoi(s);
    if (const char *s = getenv("GFFT_FUSE2_KINDS")) fuse2_kinds = atoi(s);
    if (const char *s = getenv("GFFT_FUSE2_WAIT_MS")) fuse2_wait_ms = atoi(s);
  }
};
Options &opts() {
  static Options o;
  return o;
}

// ---- twiddle tables (device resident, shared by plans) ------------------------------------
// (per device: the key's precision slot also carries the ordinal of the device the table lives on)
int dev_prec(int precision) { return precision + (current_device() << 8); }
std::mutex g_tw_mutex;
std::map<std::tuple<int64_t, int, int>, void *> g_tw_cache;       // (n, precision + device, test-hook signature) -> W_n^k, k<n
struct BigTw { void *hi, *lo; int L; };
std::map<std::pair<int64_t, int>, BigTw> g_bigtw_cache;     // (big_n, precision)

// exp(-2 pi i k / n) for k in [k0, k0 + count*step) step `step`, in long double, stored as `precision`
int upload_twiddles(int64_t n, int64_t step, int64_t count, int precision, void **out, int64_t wrong_index = -1, double wrong_by = 0) {
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
    if (j == wrong_index) c += (long double)wrong_by;          // (test hook, Options::debug_tw_exp)
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
  const int e = opts().debug_tw_exp;
  const int64_t wrong = e > 0 ? ((int64_t)opts().debug_tw_index % n + n) % n : -1;
  auto key = std::make_tuple(n, dev_prec(precision), e > 0 ? (int)(wrong * 64 + (e & 63)) + 1 : 0);
  auto it = g_tw_cache.find(key);
  if (it == g_tw_cache.end()) {
    void *d = nullptr;
    int rc = upload_twiddles(n, 1, n, precision, &d, wrong, e > 0 ? std::pow(10.0, -e) : 0.0);
    if (rc) return rc;
    it = g_tw_cache.emplace(key, d).first;
  }
  *out = it->second;
  return GFFT_OK;
}

int get_bigtw(int64_t big_n, int precision, BigTw *out) {
  std::lock_guard<std::mutex> lock(g_tw_mutex);
  auto key = std::make_pair(big_n, dev_prec(precision));
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
// Buffers a pass may read/write.  IN/OUT are the caller's arrays; the other three are regions of
// one plan-owned scratch allocation (made at the first execute):
//   WS   an array-sized workspace: the padded-pitch W of the 3-D schedule, or the complex staging
//        of a multi-axis c2r (so the caller's input is never written)
//   FS   the transposed intermediate of a four-step axis
//   AUX  the embedding buffer of a Bluestein / real-as-complex axis
//   RING the hand-off ring (+ its counters) of a fused pass pair (PK_FUSED2)
//   FS2  the intermediate of the outer level of a three-pass axis (lengths beyond 2^24, plan_long)
enum Buf { BUF_IN = 0, BUF_OUT = 1, BUF_WS = 2, BUF_FS = 3, BUF_AUX = 4, BUF_RING = 5, BUF_FS2 = 6, BUF_COUNT = 7 };

enum PassKind { PK_FFT = 0, PK_EMBED, PK_MULB, PK_EXTRACT, PK_FUSED2, PK_PLANE2D };    // PK_PLANE2D: both passes of small planes on chip (fft_plane2d.hip)

struct Pass {
  PassKind kind = PK_FFT;
  PassDesc d{};
  Factors f{};
  bool regk = false, cols = false;   // register-resident kernel family (else fft_generic); mapping
  int src = BUF_IN, dst = BUF_OUT;
  bool carries_scale = false;        // the plan's scale factor is applied by this pass
  bool logical_first = false;        // first kernel of a transformed axis (profiling/report)
  PointDesc pt{};                    // PK_EMBED / PK_MULB / PK_EXTRACT geometry
  double extra_scale = 1.0;          // e.g. 1/M of Bluestein's inverse
  // packed-real passes keep the full-length form of the same line (MODE_R2C / MODE_C2R with the
  // zero-imaginary / mirror adapters) for what only that form can fuse: truncation / padding
  bool has_full = false;
  PassDesc full{};
  int64_t data_inner = 0;            // columns of d.inner that carry data (0 = all): byte accounting
  // guru plans: blocks of the transformed axis per side and the distance between their starts
  // (gfft_plan_set_tiles re-derives the block jumps from them)
  int blocks[2] = {1, 1};
  int64_t bstride[2] = {0, 0};
  // PK_FUSED2: `d` = pass A on ONE plane, `d2` = pass B on one plane, `fused` = the planes (fft_pow2_impl.h
  // fft_fused2_kernel); fused.ctr and the ring's address are filled in at execution
  PassDesc d2{};
  FusedDesc fused{};
  int fused_kind = 0, fused_variant = 1;
  PassDesc *dev_descs = nullptr;     // {d, d2} in device memory (owned by the plan: gfft_plan_s::device_allocs)
  // the same two passes as stand-alone launches (gfft_plan_s::alt[alt_first], [alt_first + 1]): what the plan runs
  // once a fused launch has given up a wait; alt_buf / alt_bytes = the scratch region only that form needs
  int alt_first = -1;
  int alt_buf = -1;
  size_t alt_bytes = 0;
  double bytes2 = 0;                 // algorithmic bytes of pass B (gfft_plan_pass_info reports A + B)
};

// Turn two consecutive passes A -> B into one fused launch when a kernel pair exists: dA / dB are the passes
// cut down to one plane, the plane strides say where plane p starts on A's input and B's output side.
bool make_fused2(gfft_plan_s *pl, int kind, const Pass &a, const Pass &b, const PassDesc &dA, const PassDesc &dB, int planes,
                 int64_t a_in_plane, int64_t b_out_plane, int64_t slot_bytes, Pass *out);

bool is_pow2(int64_t n) { return n > 0 && (n & (n - 1)) == 0; }

/* threads per line (NT = n / R) of the power-of-two ROW kernels, as the tables in fft_pow2_f64.hip /
 * fft_pow2_f32.hip instantiate them (default variants); launch_pow2_one re-checks at launch */
static int pow2_rows_nt(int64_t n, int precision) {
  if (!is_pow2(n) || n < 16 || n > 4096) return 0;
  if (n <= 32) return 4;
  if (n == 64) return 8;
  if (n == 128) return 16;
  if (n == 256) return precision == 8 ? 32 : 16;
  if (n <= 1024) return 64;
  return (int)(n / 16);
}

// lengths served by the register-resident kernels: 2^k (16..4096), 3^b 2^k (48..3456), 5^c 2^k (20..4000)
bool regk_ok(int64_t n, int precision) {
  if (opts().force_generic) return false;
  if (n > 4096) return false;
  if (mix3_supported((int)n) || mix5_supported((int)n)) return true;
  return precision == 8 ? pow2_supported_f64((int)n) : pow2_supported_f32((int)n);
}

// ... plus the lengths that have plain COMPLEX one-pass kernels only (3 x 5 x 2^k and neighbours, fft_mixv_*.hip: no real
// modes, no fused truncation, no four-step twiddle, no exchange-buffer layouts): natural-layout c2c lines and the complex
// passes of the one-rank 3-D schedule take them; everything else keeps the paths it had
bool regk_c2c_ok(int64_t n, int precision, int mode) {
  (void)precision;
  return !opts().force_generic && opts().mixv && mode == MODE_C2C && n <= 4096 && mixv_supported((int)n);
}

// ... and packed-real rows of twice such a length (complex length m = n / 2): plain rows only
bool real_half_mixv_ok(int64_t m) { return !opts().force_generic && opts().mixv && m <= 4096 && mixv_supported((int)m); }

// lengths whose kernels carry the fused 3/2-rule truncation / zero-padding adapters: every register-kernel
// length except 5^c 2^k, whose adapters only a `make VARIANTS=1` library instantiates (fft_pow2_impl.h
// TABLE_FLAGS & 1024, kFusedPadMix5) -- plans on those lengths keep the separate gfft_truncate / gfft_pad
// kernels.  `n_axis`: transformed length of a complex axis, or the COMPLEX length (half) of a packed-real row.
bool fused_pad_ok(int64_t n_axis) {
#ifdef GFFT_VARIANTS
  // (a `make VARIANTS=1` library instantiates the 5^c 2^k adapters; the unequal-width lengths not divisible by 3 have none in ANY build)
  return !(n_axis <= 4096 && mixv_supported((int)n_axis) && n_axis % 3 != 0);
#else
  // (the unequal-width stage kernels carry the adapters on lengths divisible by 3: what the 3/2-rule makes of 5 x 2^k, 7 x 2^k ...)
  return !(n_axis <= 4096 && (mix5_supported((int)n_axis) || (mixv_supported((int)n_axis) && n_axis % 3 != 0)));
#endif
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

// the generic kernel evaluates a prime radix r by definition, O(r^2) per butterfly: fine for the
// small odd primes of everyday sizes, hopeless for big ones -> Bluestein above this
constexpr int GENERIC_MAX_PRIME = 61;

size_t align256(size_t b) { return (b + 255) & ~(size_t)255; }

// ---- scratch shared by all plans ---------------------------------------------------------------
// One buffer per (device, host thread, stream) -- torch's default stream is handle 0 on EVERY device --, grown to the
// largest request seen there: the passes of one
// gfft_execute are enqueued back to back by one thread and a stream runs them in order, so every
// plan that thread executes on that stream can share the buffer (the forward and the backward plan
// of a 1024^3 PFFT each need a 16 GiB padded workspace; owning one each doubled that).  Other
// threads -- whose launches could interleave with this one's on the same stream -- and other
// streams get their own.  Growing synchronises the stream once (earlier launches may still be
// using the old buffer).
struct ScratchPool {
  struct Slot { void *p = nullptr; size_t n = 0; bool pinned = false; };
  struct Key {
    int dev; std::thread::id th; hipStream_t st;
    bool operator<(const Key &o) const { return std::tie(dev, th, st) < std::tie(o.dev, o.th, o.st); }
  };
  std::mutex m;
  std::map<Key, Slot> bufs;
  std::vector<void *> retired;      // buffers a captured graph may still use, replaced by larger ones
  int get(hipStream_t s, size_t bytes, void **out) {
    std::lock_guard<std::mutex> lock(m);
    const auto me = std::this_thread::get_id();
    const int dev = current_device();
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cap) != hipSuccess) (void)hipGetLastError();
    if (cap != hipStreamCaptureStatusNone) {
      // No allocation inside a capture: the graph uses the buffer this thread's warm-up execution
      // grew (on whatever stream that ran), so replays must not overlap other executions of this
      // thread -- the stream a graph is replayed on orders them.  The buffer's address is now baked
      // into the graph: it is PINNED -- never freed by growth or by the last plan going away, only
      // by an explicit gfft_scratch_release().
      for (auto &kv : bufs)
        if (kv.first.dev == dev && kv.first.th == me && kv.second.n >= bytes) { kv.second.pinned = true; *out = kv.second.p; return GFFT_OK; }
      return fail(GFFT_ERR_INVALID, "stream capture: execute the plan once before capturing it (its workspace is allocated at the first execution)");
    }
    Slot &b = bufs[Key{dev, me, s}];
    if (bytes > b.n) {
      if (b.p && b.pinned) {
        retired.push_back(b.p);      // a captured graph holds this address: keep it alive
      } else if (b.p) {
        HIP_TRY(hipStreamSynchronize(s));
        HIP_TRY(hipFree(b.p));
      }
      b = Slot{};
      void *p = nullptr;
      hipError_t e = hipMalloc(&p, bytes);
      if (e == hipErrorOutOfMemory) { (void)hipGetLastError(); return fail(GFFT_ERR_NOMEM, "scratch allocation failed"); }
      HIP_TRY(e);
      b.p = p;
      b.n = bytes;
    }
    *out = b.p;
    return GFFT_OK;
  }
  // everything == false: the last plan of the process went away -- buffers no graph refers to are
  // freed; true: gfft_scratch_release(), the caller vouches that no captured graph will be replayed
  int release(bool everything) {
    std::lock_guard<std::mutex> lock(m);
    // (one device-wide synchronisation per device that holds a buffer: the stream handles in the keys may have been
    // destroyed since)
    const int here = current_device();
    int synced = -1;
    for (auto it = bufs.begin(); it != bufs.end();) {
      if (it->second.p && (everything || !it->second.pinned)) {
        if (it->first.dev != synced) {          // (keys are ordered by device)
          if (hipSetDevice(it->first.dev) != hipSuccess || hipDeviceSynchronize() != hipSuccess) (void)hipGetLastError();
          synced = it->first.dev;
        }
        (void)hipFree(it->second.p);
        it = bufs.erase(it);
      } else {
        ++it;
      }
    }
    if (synced >= 0 && synced != here && hipSetDevice(here) != hipSuccess) (void)hipGetLastError();
    if (everything) {
      if (hipDeviceSynchronize() != hipSuccess) (void)hipGetLastError();
      for (void *p : retired) (void)hipFree(p);
      retired.clear();
    }
    return GFFT_OK;
  }
};
ScratchPool &scratch_pool() {
  static ScratchPool p;
  return p;
}
// plans alive: when the last one goes the shared workspaces go with it (a 1024^3 plan leaves 16 GiB)
std::atomic<int> g_live_plans{0};

}  // namespace

struct gfft_plan_s {
  int ndims = 0, kind = 0, precision = 0;
  std::vector<int64_t> sizes_in, sizes_out;
  std::vector<int> axes;
  std::vector<Pass> passes;
  size_t region_bytes[BUF_COUNT] = {0, 0, 0, 0, 0, 0, 0};   // WS / FS / AUX / RING / FS2 sizes
  double flops = 0, bytes = 0;
  int variant_rows = 0, variant_cols = 0, xcd_swizzle = 0;
  bool fused3 = false;
  int mixv_variant = 0;                        // option mixv_variant at plan time (fft_mixv_*.hip: measured alternatives)
  int64_t ws_pitch = 0;                        // plan_fused3: entries between consecutive rows of the workspace
  int ws_tile = 0;                             // ... or: columns of a tile of its tile-major layout
  std::vector<int64_t> trunc;                  // gfft_plan_create_padded: kept entries per axis (else empty)
  std::vector<std::vector<hipEvent_t>> prof;   // per execute: events before pass 0 and after each pass
  std::vector<void *> device_allocs;            // small device buffers the plan owns (descriptors of fused launches)
  std::vector<Pass> alt;                        // stand-alone forms of the fused pairs (Pass::alt_first)
  unsigned id = 0;                              // what a voided fused launch reports (async_errors)
  int device = 0;                               // the device the plan's tables and descriptors live on
  std::atomic<bool> fused_off{false};           // a fused launch gave up a wait: the pairs run as stand-alone passes from now on
  std::atomic<bool> voided{false};              // ... and nobody has been told yet (poll_async_error)
  int async_slot = -1;                          // this plan's word of the host-visible table (async_errors; -1: none taken yet)
  gfft_plan_s();
  ~gfft_plan_s();
};

// ---- fused launches that gave up a wait (fft_pow2_impl.h fused_give_up) --------------------------------
// The kernel writes the plan's id into the plan's OWN word of a small table of pinned host words (taken at the plan's first
// fused launch, given back when the plan is destroyed: two live plans never share one); the library scans the table on entry to gfft_execute, in
// gfft_plan_status() and in gfft_async_error().  A plan named there switches its pairs to stand-alone passes and carries
// the error until it is reported ONCE: by that plan's own next gfft_execute (which enqueues nothing), by
// gfft_plan_status(plan), or by gfft_async_error() -- whichever looks first.  Other plans are not refused: the results of
// the voided plan's last execution are invalid, everything else is untouched.
namespace {
struct AsyncErrors {
  static constexpr int SLOTS = 256;
  std::mutex m;
  unsigned *flag = nullptr;                    // SLOTS pinned, device-visible words
  gfft_plan_s *owner[SLOTS] = {};              // the live plan each word belongs to (a plan takes one at its first fused launch)
  std::map<unsigned, gfft_plan_s *> live;
  std::vector<unsigned> orphans;               // ids of voided plans destroyed before anybody looked
  unsigned next_id = 1;
  unsigned *word() {
    if (!flag) {
      void *p = nullptr;
      if (hipHostMalloc(&p, SLOTS * sizeof(unsigned), hipHostMallocPortable | hipHostMallocMapped) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
      flag = static_cast<unsigned *>(p);
      for (int i = 0; i < SLOTS; ++i) flag[i] = 0;
    }
    return flag;
  }
};
AsyncErrors &async_errors() {
  static AsyncErrors a;
  return a;
}
// move what the kernels wrote since the last look into the plans (a.m held by the caller)
void collect_async_errors(AsyncErrors &a) {
  if (!a.flag) return;
  for (int i = 0; i < AsyncErrors::SLOTS; ++i) {
    // (a plain look first -- this runs on entry to every gfft_execute --, then ONE exchange, not load-then-store: a word written
    // between the two would be lost)
    if (!__atomic_load_n(a.flag + i, __ATOMIC_RELAXED)) continue;
    const unsigned id = __atomic_exchange_n(a.flag + i, 0u, __ATOMIC_ACQ_REL);
    if (!id) continue;
    auto it = a.live.find(id);
    if (it == a.live.end()) { a.orphans.push_back(id); continue; }
    gfft_plan_s *pl = it->second;
    pl->fused_off = true;
    pl->voided = true;
    for (const Pass &p : pl->passes)
      if (p.kind == PK_FUSED2 && p.alt_buf >= 0 && p.alt_bytes > pl->region_bytes[p.alt_buf]) pl->region_bytes[p.alt_buf] = p.alt_bytes;
  }
}
int report_voided(unsigned id) {
  char msg[320];
  snprintf(msg, sizeof msg, "a fused launch of plan #%u waited longer than %d ms for another workgroup and gave up (device shared or stalled?): "
           "the results of that plan's last execution are INVALID; the plan runs its pass pairs as stand-alone launches from now on", id,
           opts().fuse2_wait_ms);
  return fail(GFFT_ERR_VOIDED, msg);
}
// `only`: report this plan's pending event (if any); nullptr: report any pending event
int poll_async_error(gfft_plan_s *only) {
  AsyncErrors &a = async_errors();
  if (!a.flag) return GFFT_OK;
  std::lock_guard<std::mutex> lock(a.m);
  collect_async_errors(a);
  if (only) {
    if (!only->voided.exchange(false)) return GFFT_OK;
    return report_voided(only->id);
  }
  for (auto &kv : a.live)
    if (kv.second->voided.exchange(false)) return report_voided(kv.first);
  if (!a.orphans.empty()) {
    const unsigned id = a.orphans.back();
    a.orphans.pop_back();
    return report_voided(id);
  }
  return GFFT_OK;
}
}  // namespace
gfft_plan_s::gfft_plan_s() {
  device = current_device();
  AsyncErrors &a = async_errors();
  std::lock_guard<std::mutex> lock(a.m);
  id = a.next_id++;
  if (!a.next_id) a.next_id = 1;
  a.live[id] = this;
}
gfft_plan_s::~gfft_plan_s() {
  {
    AsyncErrors &a = async_errors();
    std::lock_guard<std::mutex> lock(a.m);
    if (a.flag && async_slot >= 0) {
      // (a launch of this plan may have voided itself since the last look: its word goes with the slot)
      if (__atomic_exchange_n(a.flag + async_slot, 0u, __ATOMIC_ACQ_REL) == id) voided = true;
      a.owner[async_slot] = nullptr;
    }
    if (voided.exchange(false)) a.orphans.push_back(id);      // destroyed before anybody looked: still reported, by gfft_async_error
    a.live.erase(id);
  }
  for (void *q : device_allocs) (void)hipFree(q);
}

namespace {

void need(gfft_plan_s *pl, int buf, size_t bytes) {
  if (bytes > pl->region_bytes[buf]) pl->region_bytes[buf] = bytes;
}

// Slots of the hand-off ring and planes the producer runs ahead (options fuse2_ring / fuse2_lag, else automatic); false:
// this launch has too few planes for a ring on which the pair pays.
// How far the producer runs ahead is a matter of BYTES, not planes: ~96 MiB of lead, twice that of ring (the Infinity
// Cache holds 256 MiB) -- 6 / 12 planes of 16 MiB (complex128, n = 1024: the optimum of the round-3 sweeps), 12 / 24
// planes of 8 MiB (complex64: with 6 / 12 the pair LOSES, 21.1 -> 21.9 ms per 1024^3 step, with 12 / 24 it gains, ->
// 18.9 ms; profiles/r04_ab_fuse2_f32.txt), 24 / 48 planes of 4 MiB (complex128 n = 512: 16 / 32 +11 %, 24 / 48 -11 %).
bool fused2_ring(int precision, int n_a, int n_b, int64_t slot_bytes, int planes, int *ring_out, int *lag_out) {
  int ring = opts().fuse2_ring, lag = opts().fuse2_lag;
  if (ring <= 0) {
    int64_t ahead = (((int64_t)96 << 20) + slot_bytes - 1) / (slot_bytes > 0 ? slot_bytes : 1);
    ahead = ahead < 4 ? 4 : (ahead > 32 ? 32 : ahead);
    ring = 2 * (int)ahead;
    // too few planes for that: the complex128 n = 1024 pairs still pay on a shorter ring (8 / 4: 34.3 against 39.9 ms
    // unfused, round 3); the others LOSE with less lead than this and stay unfused
    if (planes < 2 * ring) {
      if (!(precision == GFFT_F64 && (n_a == 1024 || n_b == 1024))) return false;      // (the real fp64 pairs: 12 / 6 ... 24 / 12 level)
      while (ring > 8 && planes < 2 * ring) ring -= 2;
    }
  }
  if (lag <= 0) lag = ring / 2;
  if (planes < 2 * ring || lag < 1 || ring <= lag) return false;
  *ring_out = ring;
  *lag_out = lag;
  return true;
}

// The hand-off protocol of the fused pairs leans on gfx94x / gfx950 specifics (make_fused2): elsewhere the pairs stay off, and
// plan_fused3 asks BEFORE it lays the workspace out for a pair it would not get.
bool fused2_arch_ok() {
  static int arch_ok[kMaxDevices] = {};          // 0 unknown, 1 yes, -1 no
  const int dev = current_device();
  if (!arch_ok[dev]) {
    hipDeviceProp_t prop;
    arch_ok[dev] = -1;
    if (hipGetDeviceProperties(&prop, dev) == hipSuccess) {
      if (!strncmp(prop.gcnArchName, "gfx942", 6) || !strncmp(prop.gcnArchName, "gfx950", 6)) arch_ok[dev] = 1;
    } else {
      (void)hipGetLastError();
    }
  }
  return arch_ok[dev] > 0;
}

bool make_fused2(gfft_plan_s *pl, int kind, const Pass &a, const Pass &b, const PassDesc &dA, const PassDesc &dB, int planes,
                 int64_t a_in_plane, int64_t b_out_plane, int64_t slot_bytes, Pass *out) {
  // auto: 12 slots with the producer 6 planes ahead where there are planes enough, else 8 / 4.  Swept on one box, plans
  // alternating on the same arrays (tools/fused2_ring_sweep.py, profiles/r03_fused2_ring_sweep_final.txt): 1024^3
  // complex128 per step 34.3 (8 / 4) -> 32.4 (12 / 6), 32.2 (14 / 7), 31.8 (13 / 6, 11 / 6); lag 3 or ring 6: 37.9;
  // C2 0.748 -> 0.710 ms (16 / 8: 0.70).  (Before the row tiles lost their barriers 8 / 4 was the optimum: a
  // faster consumer wants the producer further ahead.)
  // The hand-off protocol leans on gfx94x / gfx950 specifics: raw-buffer cache policy sc0 | sc1 = system scope (written
  // through / never served from a stale L2 line), stores counted by vmcnt -- so that s_waitcnt(0) means "write-through
  // acknowledged" --, relaxed agent-scope atomics as the only ordering.  On a target with another memory model (a
  // separate store counter, other policy bits) the pairs stay off: the stand-alone passes are always correct.
  if (!fused2_arch_ok()) return false;
  int ring = 0, lag = 0;
  if (!fused2_ring(pl->precision, dA.n, dB.n, slot_bytes, planes, &ring, &lag)) return false;
  if (!opts().fuse2 || !((opts().fuse2_kinds >> kind) & 1)) return false;
#ifdef GFFT_VARIANTS      // (make VARIANTS=1: the 8-lines-per-tile kernel sets, option fuse2 = 2 / 4 -- measured a quarter slower, fft_fused_f64.hip)
  int variant = (opts().fuse2 == 2 || opts().fuse2 == 4) ? opts().fuse2 : 1;
#else
  int variant = 1;
#endif
  if (variant == 1 && gfft::g_fuse2_n512 == 2 && pl->precision == GFFT_F64 && dA.n == 512 && dB.n == 512 && (kind == FUSED_COLS_ROWS || kind == FUSED_PLANES_CR_B))
    variant = 5;
  // (complex64 pairs are on by default since round 4 -- option fuse2_f32 = 1, profiles/r04_ab_fuse2_f32.txt; the real fp32
  // pairs measured level with their stand-alone passes and need fuse2_f32 = 2)
  const bool real_kind = kind == FUSED_R2C_PLANES || kind == FUSED_COLS_C2R;
  const bool f32 = pl->precision == GFFT_F32;
  if (f32 && (!opts().fuse2_f32 || (real_kind && opts().fuse2_f32 < 2))) return false;
  if (f32 ? (real_kind ? !fused2_real_supported_f32(kind, dA.n, dB.n) : !fused2_supported_f32(kind, dA.n, dB.n))
          : (real_kind ? !fused2_real_supported_f64(kind, dA.n, dB.n) : !fused2_supported_f64(kind, variant, dA.n, dB.n))) return false;
  int ta = 0, tb = 0;
  if ((f32 ? (real_kind ? fused2_real_tiles_f32(kind, dA, dB, &ta, &tb) : fused2_tiles_f32(kind, dA, dB, &ta, &tb))
           : (real_kind ? fused2_real_tiles_f64(kind, dA, dB, &ta, &tb) : fused2_tiles_f64(kind, variant, dA, dB, &ta, &tb))) || ta < 1 || tb < 1) return false;
  // (hand-off accesses carry 32-bit byte offsets inside a slot; tickets are 32-bit)
  if (slot_bytes >= ((int64_t)1 << 31) || (double)planes * (ta + tb) >= 1.0e9) return false;     // (tickets < 2^30: a launch that gives up pushes the counter 2^31 on)
  Pass f = a;
  f.kind = PK_FUSED2;
  f.fused_kind = kind;
  f.fused_variant = variant;
  f.d = dA;
  f.d2 = dB;
  f.src = a.src;
  f.dst = b.dst;
  f.carries_scale = false;
  f.fused.planes = planes;
  f.fused.tiles_a = ta;
  f.fused.tiles_b = tb;
  f.fused.ring = ring;
  f.fused.lag = lag;
  f.fused.defer = 0;      // an A tile's counter is settled at the end of the tile (1 / 2: behind the next tile's loads / the next ticket's poll -- measured, not kept)
  f.fused.group = 1;      // one tile per ticket (groups of 2 / 4 measured level to slower, profiles/r03_ab_fuse2_group.txt)
  f.fused.a_in_plane = a_in_plane;
  f.fused.b_out_plane = b_out_plane;
  f.fused.slot_bytes = (int64_t)align256((size_t)slot_bytes);
  f.fused.ctr = nullptr;
  f.fused.wait_ticks = 0;            // (filled in at execution: option fuse2_wait_ms)
  f.fused.plan_id = pl->id;
  f.fused.host_flag = nullptr;
  f.fused.debug = 0;
  f.alt_first = (int)pl->alt.size();
  pl->alt.push_back(a);
  pl->alt.push_back(b);
  for (const Pass *q : {&a, &b})
    for (int side : {q->src, q->dst})
      if (side >= BUF_WS && side != BUF_RING) f.alt_buf = side;     // (at most one scratch region: WS of the 3-D schedule, FS of a four-step pair)

  size_t ctr_bytes = align256((size_t)(16 + 2 * planes) * sizeof(unsigned));
#ifdef GFFT_FUSE2_TRACE
  ctr_bytes += 256 + (size_t)1024 * 96 * 16 * sizeof(unsigned long long);       // (fft_pow2_impl.h, GFFT_TRACE_STAMP)
#endif
  need(pl, BUF_RING, (size_t)ring * (size_t)f.fused.slot_bytes + ctr_bytes);
  // the kernel reads the two descriptors from device memory (the scale factors travel as kernel arguments)
  const PassDesc both[2] = {f.d, f.d2};
  void *dev = nullptr;
  if (hipMalloc(&dev, sizeof both) != hipSuccess) { (void)hipGetLastError(); return false; }
  if (hipMemcpy(dev, both, sizeof both, hipMemcpyHostToDevice) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(dev); return false; }
  pl->device_allocs.push_back(dev);
  f.dev_descs = static_cast<PassDesc *>(dev);
  *out = f;
  return true;
}

// A batch of 1-D lines: array [outer][nin -> nout][inner] (row-major), logical length n.
struct Line {
  int64_t outer, inner, n, nin, nout;
  int mode;        // PassMode
  bool inverse;
  int src, dst;
};

int plan_line(gfft_plan_s *pl, const Line &L, bool top);

PassDesc natural_desc(const Line &L) {
  PassDesc d{};
  d.n = (int)L.n;
  d.mode = L.mode;
  d.conj_in = L.inverse ? 1 : 0;
  d.conj_out = (L.inverse && L.mode != MODE_C2R) ? 1 : 0;
  d.batch = L.outer * L.inner;
  d.mid = 1;
  d.inner = L.inner;
  d.in_os = L.nin * L.inner;
  d.in_is = 1;
  d.in_es = L.inner;
  d.out_os = L.nout * L.inner;
  d.out_is = 1;
  d.out_es = L.inner;
  d.scale = 1.0;
  return d;
}

// Real lines along a contiguous axis, even length: the packed-real form (MODE_R2C_H / MODE_C2R_H).
bool real_half_ok(const Line &L, int prec) {
  (void)prec;
  return !opts().force_generic && (L.mode == MODE_R2C || L.mode == MODE_C2R) &&
         L.inner == 1 && L.n % 2 == 0 && L.n / 2 <= 4096 &&
         (real_half_supported((int)(L.n / 2)) || real_half_mix_supported((int)(L.n / 2)) || real_half_mixv_ok(L.n / 2)) && L.outer < ((int64_t)1 << 31);
}

// natural_desc(L) restated for the packed-real form: d.n = complex length, real side in pairs
void half_desc(PassDesc *d, const Line &L) {
  const int64_t m = L.n / 2;
  d->n = (int)m;
  d->mode = L.mode == MODE_R2C ? MODE_R2C_H : MODE_C2R_H;
  d->conj_in = 0;
  d->conj_out = L.mode == MODE_C2R ? 1 : 0;     // inverse by conjugation: the pre-pass conjugates
  d->in_os = L.mode == MODE_R2C ? m : m + 1;
  d->out_os = L.mode == MODE_R2C ? m + 1 : m;
  d->in_es = d->out_es = 1;
  d->in_is = d->out_is = 1;
}

// ---- four-step for long composite lengths: n = n1 * n2 (complex) ---------------------------
int plan_fourstep(gfft_plan_s *pl, const Line &L, int64_t n1, int64_t n2) {
  const int prec = pl->precision;
  const int64_t esz = 2 * prec, n = L.n, inner = L.inner, outer = L.outer;
  BigTw bt;
  int rc = get_bigtw(n, prec, &bt);
  if (rc) return rc;
  // The intermediate tmp[o][i2][k1][i] lives in the FS region with its i2-stride S2 padded off
  // the power of two (same channel-aliasing argument as the 3-D workspace).
  int64_t S2 = n1 * inner;
  if ((S2 * esz) % 2048 == 0) S2 += 256 / esz;
  Pass a, b;
  // step 1: length-n1 transforms over i1 (stride n2*inner) for every (o, i2, i); the store
  // applies W_n^(i2*k1) and writes transposed-within-row
  a.d = natural_desc(L);
  a.d.n = (int)n1;
  a.d.batch = outer * n2 * inner;
  a.d.mid = n2;
  a.d.inner = inner;
  a.d.in_os = n * inner;  a.d.in_ms = inner;  a.d.in_is = 1;  a.d.in_es = n2 * inner;
  a.d.out_os = n2 * S2;   a.d.out_ms = S2;    a.d.out_is = 1; a.d.out_es = inner;
  a.d.conj_in = L.inverse ? 1 : 0;
  a.d.conj_out = 0;
  a.d.tw_hi = bt.hi;  a.d.tw_lo = bt.lo;  a.d.tw_L = bt.L;  a.d.big_n = n;
  a.src = L.src;
  a.dst = BUF_FS;
  a.logical_first = true;
  // step 2: length-n2 transforms over i2 (stride S2) for every (o, k1, i); natural output
  b.d = natural_desc(L);
  b.d.n = (int)n2;
  b.d.batch = outer * n1 * inner;
  b.d.mid = 1;
  b.d.inner = n1 * inner;
  b.d.in_os = n2 * S2;   b.d.in_is = 1;  b.d.in_es = S2;
  b.d.out_os = n * inner; b.d.out_is = 1; b.d.out_es = n1 * inner;
  b.d.conj_in = 0;
  b.d.conj_out = L.inverse ? 1 : 0;
  b.src = BUF_FS;
  b.dst = L.dst;
  const int gmax = generic_max_n(prec);
  for (Pass *q : {&a, &b}) {
    const int64_t m = q->d.n;
    if (regk_ok(m, prec)) {
      q->regk = true;
      q->cols = true;
    } else if (!(m <= gmax && factorize(m, &q->f, GENERIC_MAX_PRIME))) {
      return fail(GFFT_ERR_UNSUPPORTED, "four-step factor not plannable");
    }
    rc = get_twiddles(m, prec, &q->d.tw);
    if (rc) return rc;
  }
  // Both passes in one persistent launch, a signal's intermediate handed over inside the Infinity Cache
  // (FUSED_FOURSTEP): plane = one signal, slot = its [n2][S2] intermediate.
  if (a.regk && b.regk && inner == 1 && outer < ((int64_t)1 << 30)) {
    Pass f;
    {
      // FUSED_FOURSTEP_ROWS: the intermediate the other way round, tmp[k1][i2] (rows of n2 entries, pitch S1): the
      // first pass stores its 16 columns i2 .. i2 + 15 side by side where they are, the second reads rows
      int64_t S1 = n2;
      if ((S1 * esz) % 2048 == 0) S1 += 256 / esz;
      PassDesc dA = a.d, dB = b.d;
      dA.batch = n2;
      dA.out_os = n1 * S1;  dA.out_ms = 1;  dA.out_is = 1;  dA.out_es = S1;
      dB.batch = n1;
      dB.mid = n1;  dB.inner = 1;
      dB.in_os = n1 * S1;  dB.in_ms = S1;  dB.in_is = 1;  dB.in_es = 1;
      dB.out_os = n;  dB.out_ms = 1;  dB.out_is = 1;  dB.out_es = n1;
      if (make_fused2(pl, FUSED_FOURSTEP_ROWS, a, b, dA, dB, (int)outer, n * esz, n * esz, n1 * S1 * esz, &f)) {
        f.alt_bytes = (size_t)outer * n2 * S2 * esz;
        pl->passes.push_back(f);
        return GFFT_OK;
      }
    }
    PassDesc dA = a.d, dB = b.d;
    dA.batch = n2;           // (o, i2, i) with o = 0
    dB.batch = n1;
    if (make_fused2(pl, FUSED_FOURSTEP, a, b, dA, dB, (int)outer, n * esz, n * esz, n2 * S2 * esz, &f)) {
      f.alt_bytes = (size_t)outer * n2 * S2 * esz;
      pl->passes.push_back(f);
      return GFFT_OK;
    }
  }
  need(pl, BUF_FS, (size_t)outer * n2 * S2 * esz);
  pl->passes.push_back(a);
  pl->passes.push_back(b);
  return GFFT_OK;
}

// Pick n1*n2 = n with both factors plannable in one pass; prefers the register-kernel sizes.
bool split_fourstep(int64_t n, int prec, int64_t *n1, int64_t *n2) {
  const int gmax = generic_max_n(prec);
  if (n > ((int64_t)1 << 24)) return false;
  if (is_pow2(n)) {
    int lg = 0;
    while (((int64_t)1 << lg) < n) ++lg;
    *n1 = (int64_t)1 << ((lg + 1) / 2);
    *n2 = n / *n1;
    return *n1 <= 4096;
  }
  int64_t best = 0;
  for (int64_t a = (int64_t)std::sqrt((double)n); a >= 2; --a) {
    if (n % a) continue;
    const int64_t b = n / a;
    Factors fa, fb;
    const bool oka = regk_ok(a, prec) || (a <= gmax && factorize(a, &fa, GENERIC_MAX_PRIME));
    const bool okb = regk_ok(b, prec) || (b <= gmax && factorize(b, &fb, GENERIC_MAX_PRIME));
    if (!oka || !okb) continue;
    if (regk_ok(a, prec) && regk_ok(b, prec)) { best = a; break; }
    if (!best) best = a;
  }
  if (!best) return false;
  *n1 = n / best;
  *n2 = best;
  return true;
}

// ---- lengths beyond 2^24: n = n1 * n2 with n2 a four-step (or single-pass) length again ------------------------------
// FFTW plans any length (/root/reference/mpi4py_fft/fftw/fftw_planxfftn.c:52-75).  One more level of the same
// decomposition: the strided pass of length n1 with the twiddle W_n^(i2 k1) on its store leaves tmp[o][i2][k1][i] --
// which IS the natural layout [outer][n2][inner'] of a batch of lines of length n2 with inner' = n1 * inner, and the
// result of transforming those lines in place of i2, out[o][k2][k1][i], is the natural order k = k2 n1 + k1 of the long
// transform.  So the rest is plan_line on that batch (a four-step transform with inner > 1: two passes through FS):
// three HBM round trips in all, the intermediate of this level in its own region (FS2).  Inverse: conj . forward . conj
// level by level -- this pass conjugates on load AND on store, the inner line conjugates again on its load.
int plan_long(gfft_plan_s *pl, const Line &L, int64_t n1, int64_t n2) {
  const int prec = pl->precision;
  const int64_t esz = 2 * prec, n = L.n, inner = L.inner, outer = L.outer;
  BigTw bt;
  int rc = get_bigtw(n, prec, &bt);
  if (rc) return rc;
  const int64_t S2 = n1 * inner;
  Pass a;
  a.d = natural_desc(L);
  a.d.n = (int)n1;
  a.d.batch = outer * n2 * inner;
  a.d.mid = n2;
  a.d.inner = inner;
  a.d.in_os = n * inner;  a.d.in_ms = inner;  a.d.in_is = 1;  a.d.in_es = n2 * inner;
  a.d.out_os = n2 * S2;   a.d.out_ms = S2;    a.d.out_is = 1; a.d.out_es = inner;
  a.d.conj_in = L.inverse ? 1 : 0;
  a.d.conj_out = L.inverse ? 1 : 0;
  a.d.tw_hi = bt.hi;  a.d.tw_lo = bt.lo;  a.d.tw_L = bt.L;  a.d.big_n = n;
  a.src = L.src;
  a.dst = BUF_FS2;
  a.logical_first = true;
  a.regk = true;
  a.cols = true;
  rc = get_twiddles(n1, prec, &a.d.tw);
  if (rc) return rc;
  need(pl, BUF_FS2, (size_t)outer * n * inner * esz);
  pl->passes.push_back(a);
  const Line rest{outer, S2, n2, n2, n2, MODE_C2C, L.inverse, BUF_FS2, L.dst};
  return plan_line(pl, rest, false);
}

// n1 for plan_long: a register-kernel length whose cofactor plan_line takes in one or two passes; powers of two split
// three ways as evenly as the tables allow (2^30 = 1024 x (1024 x 1024))
bool split_long(int64_t n, int prec, int64_t *n1) {
  if (n <= ((int64_t)1 << 24) || n >= ((int64_t)1 << 31)) return false;
  int64_t t1 = 0, t2 = 0;
  if (is_pow2(n)) {
    int lg = 0;
    while (((int64_t)1 << lg) < n) ++lg;
    *n1 = (int64_t)1 << (lg - 2 * (lg / 3));
    return true;
  }
  for (int64_t a = 4096; a >= 16; --a) {
    if (n % a || !regk_ok(a, prec)) continue;
    const int64_t b = n / a;
    if (b > ((int64_t)1 << 24)) break;
    if (regk_ok(b, prec) || split_fourstep(b, prec, &t1, &t2)) { *n1 = a; return true; }
  }
  return false;
}

// ---- embedding fallbacks: Bluestein, and real transforms beyond the single-pass limit --------
// Both copy the lines into a complex AUX array [outer][Lw][inner], transform there, and extract.
//   real-as-complex: Lw = n, the complex engine is whatever plan_line picks for length n
//   Bluestein      : Lw = M = 2^k >= 2n-1; x_j w_j -> FFT_M -> * B -> IFFT_M -> * w_k / M,
//                    w_j = exp(-i pi j^2 / n), B = FFT_M(conj chirp, wrapped)
std::map<std::pair<int64_t, int>, std::pair<void *, void *>> g_blue_cache;   // (n, prec) -> (chirp, B)

void host_fft_pow2(std::vector<long double> &re, std::vector<long double> &im) {
  const size_t n = re.size();
  for (size_t i = 1, j = 0; i < n; ++i) {
    size_t bit = n >> 1;
    for (; j & bit; bit >>= 1) j ^= bit;
    j ^= bit;
    if (i < j) { std::swap(re[i], re[j]); std::swap(im[i], im[j]); }
  }
  const long double PI = 3.14159265358979323846264338327950288L;
  for (size_t len = 2; len <= n; len <<= 1) {
    for (size_t k = 0; k < len / 2; ++k) {
      const long double a = -2.0L * PI * (long double)k / (long double)len;
      const long double wr = cosl(a), wi = sinl(a);
      for (size_t i = k; i < n; i += len) {
        const size_t j = i + len / 2;
        const long double tr = re[j] * wr - im[j] * wi, ti = re[j] * wi + im[j] * wr;
        re[j] = re[i] - tr; im[j] = im[i] - ti;
        re[i] += tr;        im[i] += ti;
      }
    }
  }
}

int get_bluestein(int64_t n, int64_t M, int prec, const void **chirp, const void **B) {
  std::lock_guard<std::mutex> lock(g_tw_mutex);
  auto key = std::make_pair(n, dev_prec(prec));
  auto it = g_blue_cache.find(key);
  if (it == g_blue_cache.end()) {
    const long double PI = 3.14159265358979323846264338327950288L;
    std::vector<long double> cr(n), ci(n), br(M, 0.0L), bi(M, 0.0L);
    for (int64_t j = 0; j < n; ++j) {
      const int64_t q = (j * j) % (2 * n);                  // exact phase reduction
      const long double a = PI * (long double)q / (long double)n;
      cr[j] = cosl(a);
      ci[j] = -sinl(a);                                      // w_j = exp(-i pi j^2 / n)
      br[j] = cr[j];
      bi[j] = -ci[j];                                        // b_j = conj(w_j)
      if (j) { br[M - j] = br[j]; bi[M - j] = bi[j]; }
    }
    host_fft_pow2(br, bi);
    auto upload = [&](const std::vector<long double> &xr, const std::vector<long double> &xi, void **out) -> int {
      std::vector<unsigned char> h(xr.size() * 2 * prec);
      for (size_t j = 0; j < xr.size(); ++j) {
        if (prec == 8) { ((double *)h.data())[2 * j] = (double)xr[j]; ((double *)h.data())[2 * j + 1] = (double)xi[j]; }
        else { ((float *)h.data())[2 * j] = (float)xr[j]; ((float *)h.data())[2 * j + 1] = (float)xi[j]; }
      }
      HIP_TRY(hipMalloc(out, h.size()));
      HIP_TRY(hipMemcpy(*out, h.data(), h.size(), hipMemcpyHostToDevice));
      return GFFT_OK;
    };
    void *dc = nullptr, *db = nullptr;
    int rc = upload(cr, ci, &dc);
    if (rc) return rc;
    rc = upload(br, bi, &db);
    if (rc) return rc;
    it = g_blue_cache.emplace(key, std::make_pair(dc, db)).first;
  }
  *chirp = it->second.first;
  *B = it->second.second;
  return GFFT_OK;
}

int plan_embedded(gfft_plan_s *pl, const Line &L, bool bluestein) {
  const int prec = pl->precision;
  const int64_t esz = 2 * prec;
  int64_t Lw = L.n;
  const void *chirp = nullptr, *B = nullptr;
  if (bluestein) {
    Lw = 1;
    while (Lw < 2 * L.n - 1) Lw <<= 1;
    if (Lw > ((int64_t)1 << 26)) return fail(GFFT_ERR_UNSUPPORTED, "transform length too large for Bluestein");
    int rc = get_bluestein(L.n, Lw, prec, &chirp, &B);
    if (rc) return rc;
  }
  need(pl, BUF_AUX, (size_t)L.outer * Lw * L.inner * esz);
  PointDesc pt{};
  pt.outer = L.outer;
  pt.inner = L.inner;
  pt.n = L.n;
  pt.nin = L.nin;
  pt.nout = L.nout;
  pt.Lw = Lw;
  pt.mode = L.mode;
  pt.chirp = chirp;
  pt.B = B;
  Pass e;
  e.kind = PK_EMBED;
  e.pt = pt;
  // inverse transform = conj . forward . conj: the embed conjugates its input.  With Bluestein the
  // forward machinery runs in between; without it the complex engine does the inverse itself.
  e.pt.conj = (bluestein && L.inverse) ? 1 : 0;
  e.src = L.src;
  e.dst = BUF_AUX;
  e.logical_first = true;
  pl->passes.push_back(e);
  Line C{L.outer, L.inner, Lw, Lw, Lw, MODE_C2C, false, BUF_AUX, BUF_AUX};
  int rc;
  if (bluestein) {
    rc = plan_line(pl, C, false);
    if (rc) return rc;
    Pass m;
    m.kind = PK_MULB;
    m.pt = pt;
    m.src = m.dst = BUF_AUX;
    pl->passes.push_back(m);
    C.inverse = true;
    rc = plan_line(pl, C, false);
    if (rc) return rc;
  } else {
    C.inverse = L.inverse;
    rc = plan_line(pl, C, false);
    if (rc) return rc;
  }
  Pass x;
  x.kind = PK_EXTRACT;
  x.pt = pt;
  x.pt.conj = (bluestein && L.inverse) ? 1 : 0;
  x.extra_scale = bluestein ? 1.0 / (double)Lw : 1.0;
  x.src = BUF_AUX;
  x.dst = L.dst;
  pl->passes.push_back(x);
  return GFFT_OK;
}

// ---- real-to-real kinds (DCT / DST I-IV; FFTW's REDFTxx / RODFTxx, fftw_planxfftn.c:68-75) ------
// Every kind is one complex DFT of its logical length N (2(n-1), 2n or 2(n+1): the figure
// get_normalization uses, xfftn.py:763-806) between two pointwise passes:
//   z[pos0 + j] = pre[j] x[j]  (zeros elsewhere)  ->  Z = DFT_N z  ->  y[k] = Re(post[k] Z[idx0 + k])
// with (theta = pi/(2n))
//   REDFT00  N=2(n-1)  pre = 1,2,...,2,1                         post = 1
//   REDFT10  N=2n      pre = 1                                   post = 2 e^{-i k theta}
//   REDFT01  N=2n      pre = (1,2,2,...) e^{-i j theta}          post = 1
//   REDFT11  N=2n      pre = e^{-i j theta}                      post = 2 e^{-i (2k+1) theta/2}
//   RODFT00  N=2(n+1)  pre = 1, pos0 = 1                         post = 2i, idx0 = 1
//   RODFT10  N=2n      pre = 1                                   post = 2i e^{-i (k+1) theta}, idx0 = 1
//   RODFT01  N=2n      pre = i (2,...,2,1) e^{-i (j+1) theta}, pos0 = 1      post = 1
//   RODFT11  N=2n      pre = e^{-i j theta}                      post = 2i e^{-i (2k+1) theta/2}
// (cos/sin sums written as real parts of complex exponentials).  The complex transform runs on
// whatever plan_line picks for N, so every length works; the cost is that of a length-N complex
// transform per n real samples -- these kinds are a convenience of the API, not its hot path.
struct R2RTables { void *pre, *post; };
std::map<std::tuple<int, int64_t, int>, R2RTables> g_r2r_cache;

int64_t r2r_logical_n(int kind, int64_t n) {
  if (kind == GFFT_REDFT00) return 2 * (n - 1);
  if (kind == GFFT_RODFT00) return 2 * (n + 1);
  return 2 * n;
}

int get_r2r_tables(int kind, int64_t n, int prec, R2RTables *out) {
  std::lock_guard<std::mutex> lock(g_tw_mutex);
  auto key = std::make_tuple(kind, n, dev_prec(prec));
  auto it = g_r2r_cache.find(key);
  if (it == g_r2r_cache.end()) {
    const long double PI = 3.14159265358979323846264338327950288L;
    const long double th = PI / (2.0L * (long double)n);
    std::vector<long double> ar(n), ai(n), br(n), bi(n);
    auto cis = [](long double a, long double *re, long double *im) { *re = cosl(a); *im = -sinl(a); };   // e^{-ia}
    for (int64_t j = 0; j < n; ++j) {
      long double pr = 1, pi_ = 0, qr = 1, qi = 0, c, sre, sim;
      switch (kind) {
        case GFFT_REDFT00: pr = (j == 0 || j == n - 1) ? 1 : 2; break;
        case GFFT_REDFT10: cis(th * j, &sre, &sim); qr = 2 * sre; qi = 2 * sim; break;
        case GFFT_REDFT01: c = j == 0 ? 1 : 2; cis(th * j, &sre, &sim); pr = c * sre; pi_ = c * sim; break;
        case GFFT_REDFT11:
          cis(th * j, &pr, &pi_);
          cis(th * (2 * j + 1) / 2, &sre, &sim); qr = 2 * sre; qi = 2 * sim; break;
        case GFFT_RODFT00: qr = 0; qi = 2; break;
        case GFFT_RODFT10: cis(th * (j + 1), &sre, &sim); qr = -2 * sim; qi = 2 * sre; break;      // 2i (sre + i sim)
        case GFFT_RODFT01:
          c = j == n - 1 ? 1 : 2; cis(th * (j + 1), &sre, &sim); pr = -c * sim; pi_ = c * sre; break;   // i c (..)
        case GFFT_RODFT11:
          cis(th * j, &pr, &pi_);
          cis(th * (2 * j + 1) / 2, &sre, &sim); qr = -2 * sim; qi = 2 * sre; break;
      }
      ar[j] = pr; ai[j] = pi_; br[j] = qr; bi[j] = qi;
    }
    auto upload = [&](const std::vector<long double> &xr, const std::vector<long double> &xi, void **dst) -> int {
      std::vector<unsigned char> h(xr.size() * 2 * prec);
      for (size_t j = 0; j < xr.size(); ++j) {
        if (prec == 8) { ((double *)h.data())[2 * j] = (double)xr[j]; ((double *)h.data())[2 * j + 1] = (double)xi[j]; }
        else { ((float *)h.data())[2 * j] = (float)xr[j]; ((float *)h.data())[2 * j + 1] = (float)xi[j]; }
      }
      HIP_TRY(hipMalloc(dst, h.size()));
      HIP_TRY(hipMemcpy(*dst, h.data(), h.size(), hipMemcpyHostToDevice));
      return GFFT_OK;
    };
    R2RTables t{nullptr, nullptr};
    int rc = upload(ar, ai, &t.pre);
    if (rc) return rc;
    rc = upload(br, bi, &t.post);
    if (rc) return rc;
    it = g_r2r_cache.emplace(key, t).first;
  }
  *out = it->second;
  return GFFT_OK;
}

// one real-to-real axis: [outer][n][inner] real -> same shape; src/dst in {BUF_IN, BUF_OUT}
int plan_r2r_line(gfft_plan_s *pl, int64_t outer, int64_t n, int64_t inner, int kind, int src, int dst) {
  const int prec = pl->precision;
  if (kind == GFFT_REDFT00 && n < 2) return fail(GFFT_ERR_INVALID, "REDFT00 needs at least two points");
  const int64_t N = r2r_logical_n(kind, n);
  R2RTables t;
  int rc = get_r2r_tables(kind, n, prec, &t);
  if (rc) return rc;
  const double lines = (double)outer * (double)inner;
  pl->flops += 2.5 * (double)N * std::log2((double)(N > 1 ? N : 2)) * lines;
  pl->bytes += lines * 2.0 * (double)n * prec;
  if ((pow2_r2r_supported((int)N) || (N <= 4096 && (mix3_supported((int)N) || mix5_supported((int)N)))) &&
      !opts().force_generic && outer * inner < ((int64_t)1 << 31)) {
    // one register-kernel pass: the pointwise steps are load / store adapters of the length-N
    // complex transform (MODE_R2R), so the line is read once and written once
    Pass p;
    p.regk = true;
    p.cols = inner > 1;
    p.logical_first = true;
    p.src = src;
    p.dst = dst;
    PassDesc &d = p.d;
    d.n = (int)N;
    d.mode = MODE_R2R;
    d.batch = outer * inner;
    d.mid = 1;
    d.inner = inner;
    d.in_os = d.out_os = n * inner;
    d.in_is = d.out_is = 1;
    d.in_es = d.out_es = inner;
    d.scale = 1.0;
    d.r2r_pre = t.pre;
    d.r2r_post = t.post;
    d.r2r_n = (int)n;
    d.r2r_pos0 = (kind == GFFT_RODFT00 || kind == GFFT_RODFT01) ? 1 : 0;
    d.r2r_idx0 = (kind == GFFT_RODFT00 || kind == GFFT_RODFT10) ? 1 : 0;
    rc = get_twiddles(N, prec, &d.tw);
    if (rc) return rc;
    pl->passes.push_back(p);
    return GFFT_OK;
  }
  need(pl, BUF_WS, (size_t)outer * N * inner * 2 * prec);
  PointDesc pt{};
  pt.outer = outer;
  pt.inner = inner;
  pt.n = n;
  pt.nin = pt.nout = n;
  pt.Lw = N;
  pt.mode = MODE_R2R;
  pt.pre = t.pre;
  pt.post = t.post;
  pt.pos0 = (kind == GFFT_RODFT00 || kind == GFFT_RODFT01) ? 1 : 0;
  pt.idx0 = (kind == GFFT_RODFT00 || kind == GFFT_RODFT10) ? 1 : 0;
  Pass e;
  e.kind = PK_EMBED;
  e.pt = pt;
  e.src = src;
  e.dst = BUF_WS;
  e.logical_first = true;
  pl->passes.push_back(e);
  Line C{outer, inner, N, N, N, MODE_C2C, false, BUF_WS, BUF_WS};
  rc = plan_line(pl, C, false);
  if (rc) return rc;
  Pass x;
  x.kind = PK_EXTRACT;
  x.pt = pt;
  x.src = BUF_WS;
  x.dst = dst;
  pl->passes.push_back(x);
  return GFFT_OK;
}

int64_t max_prime_factor(int64_t n) {
  int64_t m = 1;
  for (int64_t p = 2; p * p <= n; ++p)
    while (n % p == 0) { m = p; n /= p; }
  return n > 1 ? n : m;
}

// Build the kernel passes of one transformed axis.
int plan_line(gfft_plan_s *pl, const Line &L, bool top) {
  const int prec = pl->precision;
  const int64_t n = L.n;
  const int64_t batch = L.outer * L.inner;
  if (batch >= ((int64_t)1 << 31) || n >= ((int64_t)1 << 31))
    return fail(GFFT_ERR_UNSUPPORTED, "batch or length exceeds 2^31");
  if (top) {
    const double lines = (double)batch;
    if (n > 1) pl->flops += (L.mode == MODE_C2C ? 1.0 : 0.5) * 5.0 * (double)n * std::log2((double)n) * lines;
    const double esz_in = (L.mode == MODE_R2C) ? prec : 2.0 * prec;
    const double esz_out = (L.mode == MODE_C2R) ? prec : 2.0 * prec;
    pl->bytes += lines * ((double)L.nin * esz_in + (double)L.nout * esz_out);
  }
  Pass p;
  p.d = natural_desc(L);
  p.src = L.src;
  p.dst = L.dst;
  p.logical_first = true;
  const int gmax = generic_max_n(prec);
  if (real_half_ok(L, prec)) {
    // contiguous real lines of even length: one complex transform of HALF the length on the line
    // read / written as complex pairs, Hermitian pass in registers (fft_real_*.hip)
    if (regk_ok(n, prec)) {
      p.has_full = true;
      p.full = p.d;
      int rc = get_twiddles(n, prec, &p.full.tw);
      if (rc) return rc;
    }
    half_desc(&p.d, L);
    p.regk = true;
    p.cols = false;
    int rc = get_twiddles(n / 2, prec, &p.d.tw);
    if (!rc) rc = get_twiddles(n, prec, &p.d.rtw);
    if (rc) return rc;
    pl->passes.push_back(p);
    return GFFT_OK;
  }
  if (regk_ok(n, prec) || regk_c2c_ok(n, prec, L.mode)) {
    p.regk = true;
    p.cols = L.inner > 1;
    int rc = get_twiddles(n, prec, &p.d.tw);
    if (rc) return rc;
    pl->passes.push_back(p);
    return GFFT_OK;
  }
  int64_t n1 = 0, n2 = 0;
  // 2^a 3^b 5^c lengths the single-pass tables miss (960 = 48 x 20, 1920 = 96 x 20, ...): two
  // register-kernel passes (~2.5 TB/s effective) beat the LDS generic kernel (0.5-1.5 TB/s)
  // ... and lengths with one more small prime (896 = 128 x 7, 1792 = 256 x 7, 1408 = 128 x 11):
  // a register-kernel pass plus a tiny generic pass.
  if (L.mode == MODE_C2C && n >= 240 && !opts().force_generic) {
    int64_t both = 0, one = 0;
    // a single prime factor 7 / 11 / 13 beside a register-kernel length: that pass is one column
    // per thread (fft_generic.hip, tiny_dft_kernel) and runs at strided-copy speed
    for (int64_t a : {7, 11, 13})
      if (n % a == 0 && regk_ok(n / a, prec)) one = a;
    for (int64_t a = (int64_t)std::sqrt((double)n); a >= 2; --a) {
      if (n % a) continue;
      const int64_t b = n / a;
      if (a >= 16 && regk_ok(a, prec) && regk_ok(b, prec)) { both = a; break; }
      Factors fa;
      if (!one && a <= 61 && regk_ok(b, prec) && factorize(a, &fa, GENERIC_MAX_PRIME)) one = a;
    }
    if (both) return plan_fourstep(pl, L, n / both, both);
    if (one) return plan_fourstep(pl, L, n / one, one);
  }
  if (n <= gmax && factorize(n, &p.f, GENERIC_MAX_PRIME)) {
    int rc = get_twiddles(n, prec, &p.d.tw);
    if (rc) return rc;
    pl->passes.push_back(p);
    return GFFT_OK;
  }
  if (max_prime_factor(n) <= GENERIC_MAX_PRIME && split_long(n, prec, &n1)) {
    if (L.mode == MODE_C2C) return plan_long(pl, L, n1, n / n1);
    return plan_embedded(pl, L, false);                                       // real: as a complex line of the same length
  }
  const bool splittable = max_prime_factor(n) <= GENERIC_MAX_PRIME && split_fourstep(n, prec, &n1, &n2);
  if (L.mode == MODE_C2C && splittable) return plan_fourstep(pl, L, n1, n2);
  if (L.mode != MODE_C2C && splittable) return plan_embedded(pl, L, false);   // real, long, composite
  return plan_embedded(pl, L, true);                                          // Bluestein
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
    if (!regk_ok(full[i], pl->precision) && !(pl->trunc.empty() && ((!real || i < 2) ? regk_c2c_ok(full[i], pl->precision, MODE_C2C)
                                                                                       : (full[i] % 2 == 0 && real_half_mixv_ok(full[i] / 2))))) return false;
  const int64_t bytes = full[0] * full[1] * full[2] * (real ? 1 : 2) * pl->precision;
  return bytes >= opts().fused3_min_bytes;
}

int plan_fused3(gfft_plan_s *pl) {
  const int prec = pl->precision;
  const bool real = pl->kind == GFFT_R2C || pl->kind == GFFT_C2R;
  const bool inverse = pl->kind == GFFT_C2C_BACKWARD || pl->kind == GFFT_C2R;
  // the transformed (padded, physical-side) lengths; with gfft_plan_create_padded the spectral side
  // keeps t0 x t1 x nc entries of them (3/2-rule truncation / zero padding fused into every pass)
  const std::vector<int64_t> &full = inverse ? pl->sizes_out : pl->sizes_in;
  const int64_t n0 = full[0], n1 = full[1], n2 = full[2];
  const bool tr = pl->trunc.size() == 3;
  const int64_t nc_full = real ? n2 / 2 + 1 : n2;      // complex entries per transformed row
  const int64_t nc = tr ? pl->trunc[2] : nc_full;      // complex entries per stored row
  const int64_t t0 = tr ? pl->trunc[0] : n0, t1 = tr ? pl->trunc[1] : n1;
  auto keep = [&](Pass &p, int64_t kept, int64_t of) {   // libfft.py:263-311 as the pass's store / load adapter
    if (kept == of) return;
    p.d.tr_dir = inverse ? 2 : 1;
    p.d.tr_n = p.d.tr_N = (int)kept;
    p.d.tr_even = (kept % 2 == 0) ? 1 : 0;
  };
  const int64_t esz = 2 * prec;
  // workspace row pitch: rows start on 128-B lines; +256 B when the pitch would be a multiple of 2 KiB
  const int64_t seg = 128 / esz;      // (R4: rows in whole 256-byte pieces instead measured +1 % on the real fp64 schedule: more padding columns, nothing gained)
  int64_t P = (nc + seg - 1) / seg * seg;
  if ((P * esz) % 2048 == 0) P += 256 / esz;
  // ... and never 129 x 2^k entries: the far stride of the workspace is a power of two times P, and with P = 129 x 2^k
  // the rows a far-axis tile walks lie k (2^7 + 1) 2^j bytes apart -- the one multiplier found so far that the address
  // hash of the memory channels folds onto itself (row k and row k + 2^7 j share their channel).  Measured (round 5,
  // tools/ab_combo_probe.py ws_plane_skew, profiles/r05_ab_pitch129.txt), far-axis pass alone, same arrays:
  // (1024,1024,2048) r2c f64 [1025-wide rows, P = 1032] 11.6 -> 7.2 ms; (512,1024,2048) c128 [P = 2064] 14.0 -> 6.4 ms;
  // (2048,512,2048) r2c f64 18.0 -> 7.8 ms; (1024,1024,4096) r2c f32 [2049-wide, P = 2064] 14.0 -> 8.5 ms; pitches of
  // 17 / 33 / 65 / 257 x 2^k entries (every other BASELINE-sized shape) are level with or without a skew.
  {
    auto odd = [](int64_t x) { while (x && !(x & 1)) x >>= 1; return x; };
    while (odd(P) == 129 || (P * esz) % 2048 == 0) P += seg;
  }
  need(pl, BUF_WS, (size_t)(n0 * n1 * P * esz));
  pl->ws_pitch = P;
  // columns the in-workspace passes run over: nc rounded up to whole 128-byte lines (the padding
  // columns hold zeros written by the pass that fills W), see PassDesc::inner_ld / inner_st
  const int64_t Pu = (nc + seg - 1) / seg * seg;

  auto base = [&](int n, int mode) {
    Pass p;
    p.regk = true;
    p.logical_first = true;
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
  int64_t w_i0 = P, w_i1 = n0 * P;
  // tile-major workspace (option wtile, decided below once the pair is known): W[i0][tile of TWc columns][k1][TWc], planes w_i0 apart
  bool wtile = false;
  int wt_lg = 0;
  int64_t TWc = 0, wtS = 0;          // columns of a tile; entries from one tile to the next
  // Forward r2c with rows that are not whole lines wide (odd n/2 + 1): the pass that writes the
  // caller's array would store 256-byte segments off the line grid (4.6-5.0 ms per 1024^3 pass
  // against 3.4 aligned).  Stores hurt more than loads (tools/flat_probe.py), so that pass becomes
  // the FAR-axis one with tiles over the flattened (i1, c) index of the output -- aligned stores
  // whenever n1 * (n/2+1) * 16 B is a multiple of 128 -- reading the workspace, now W[i0][i1][c],
  // through per-lane (row, column) addresses (misaligned loads, 3.8 ms).  Backward keeps the
  // layout above: its first pass already has the misalignment on the load side.
  const bool flat_out = real && !inverse && !tr && Pu != nc && (n1 * nc * esz) % 128 == 0 && opts().flat_out;
  if (flat_out) { w_i0 = n1 * P; w_i1 = P; }
  // rows: transform along axis 2.  The batch runs fastest along whichever of (i0, i1) makes the
  // rows it READS consecutive in memory (reading scattered 16-KiB rows measured 6.7 ms per pass,
  // writing them scattered 5.8 ms): forward reads the natural array -> i1 fastest; backward
  // reads W[i1][i0][c] -> i0 fastest.  Strides are in elements of each side's own type.
  auto rows = [&](int mode, bool in_ws, bool out_ws, int src, int dst) {
    Pass p = base((int)n2, mode);
    p.cols = false;
    p.d.batch = n0 * n1;
    const int64_t nat_in = n2, nat_out = n2;   // the natural side of a row pass is the physical array
    const int64_t in_i0 = in_ws ? w_i0 : n1 * nat_in, in_i1 = in_ws ? w_i1 : nat_in;
    const int64_t out_i0 = out_ws ? w_i0 : n1 * nat_out, out_i1 = out_ws ? w_i1 : nat_out;
    if (!in_ws) {   // o = i0, i = i1
      p.d.inner = n1;
      p.d.in_os = in_i0;   p.d.in_is = in_i1;
      p.d.out_os = out_i0; p.d.out_is = out_i1;
    } else {        // o = i1, i = i0
      p.d.inner = n0;
      p.d.in_os = in_i1;   p.d.in_is = in_i0;
      p.d.out_os = out_i1; p.d.out_is = out_i0;
    }
    p.d.in_es = 1;
    p.d.out_es = 1;
    if (wtile && in_ws) { p.d.in_os = TWc; p.d.in_tlg = wt_lg; p.d.in_tS = wtS; }          // (o = k1: one row of a tile; the line itself tile-major)
    if (wtile && out_ws) { p.d.out_os = TWc; p.d.out_tlg = wt_lg; p.d.out_tS = wtS; }
    p.src = src; p.dst = dst;
    if (mode != MODE_C2C && n2 % 2 == 0 &&
        (real_half_supported((int)(n2 / 2)) || real_half_mix_supported((int)(n2 / 2)) || real_half_mixv_ok(n2 / 2))) {
      // packed-real form: complex length n2/2, the real side (the user's natural array) in pairs
      p.d.n = (int)(n2 / 2);
      p.d.mode = mode == MODE_R2C ? MODE_R2C_H : MODE_C2R_H;
      p.d.conj_in = 0;
      p.d.conj_out = mode == MODE_C2R ? 1 : 0;
      if (mode == MODE_R2C) { p.d.in_os /= 2; p.d.in_is /= 2; }
      else { p.d.out_os /= 2; p.d.out_is /= 2; }
      if (mode == MODE_R2C && out_ws) p.d.out_pad = (int)(Pu - nc);
    }
    keep(p, nc, nc_full);
    return p;
  };
  // axis 1: batch (o = i0, i = c)
  auto axis1 = [&](bool in_ws, bool out_ws, int src, int dst) {
    Pass p = base((int)n1, MODE_C2C);
    p.cols = true;
    // tiles cover the line-rounded width; the natural side is masked to its nc columns
    p.d.batch = t0 * Pu;
    p.d.inner = Pu;
    p.data_inner = nc;
    if (Pu != nc) {
      if (!in_ws) p.d.inner_ld = nc;
      if (!out_ws) p.d.inner_st = nc;
    }
    p.d.in_os = in_ws ? w_i0 : t1 * nc;   p.d.in_es = in_ws ? w_i1 : nc;
    p.d.out_os = out_ws ? w_i0 : t1 * nc; p.d.out_es = out_ws ? w_i1 : nc;
    if (wtile && in_ws) { p.d.in_es = TWc; p.d.in_ilg = wt_lg; p.d.in_iS = wtS; }
    if (wtile && out_ws) { p.d.out_es = TWc; p.d.out_ilg = wt_lg; p.d.out_iS = wtS; }      // every tile one contiguous run
    p.src = src; p.dst = dst;
    keep(p, t1, n1);
    return p;
  };
  // axis 0 inside the workspace: batch (o = i1, i = c)
  auto axis0 = [&](int src, int dst) {
    Pass p = base((int)n0, MODE_C2C);
    p.cols = true;
    p.d.batch = n1 * Pu;
    p.d.inner = Pu;
    p.data_inner = nc;
    p.d.in_os = w_i1;  p.d.in_es = w_i0;
    p.d.out_os = w_i1; p.d.out_es = w_i0;
    if (wtile) {          // k1 advances by one row of a tile, the columns are tile-major
      p.d.in_os = p.d.out_os = TWc;
      p.d.in_ilg = p.d.out_ilg = wt_lg;
      p.d.in_iS = p.d.out_iS = wtS;
    }
    p.src = src; p.dst = dst;
    keep(p, t0, n0);      // in place: a column is loaded whole before its first entry is stored
    return p;
  };
  // axis 0 from W[i0][i1][c] to the natural output, tiles over the flattened (i1, c) index
  auto axis0_flat = [&](int src, int dst) {
    Pass p = base((int)n0, MODE_C2C);
    p.cols = true;
    p.d.batch = n1 * nc;
    p.d.mid = n1;
    p.d.inner = nc;
    p.d.flat = 1;
    p.d.in_os = 0;  p.d.in_ms = w_i1;  p.d.in_es = w_i0;
    p.d.out_os = 0; p.d.out_ms = nc;   p.d.out_es = n1 * nc;
    p.src = src; p.dst = dst;
    return p;
  };
  // A complex transform may run its axes in either order.  Where the last two passes can be fused into one
  // launch (make_fused2, FUSED_COLS_ROWS) the forward direction takes the backward schedule's order -- axis 1,
  // then [axis 0 -> rows] plane by plane: the fused pair then WRITES the caller's rows scattered (plane i1 =
  // rows 16 MiB apart), which costs nothing, where the mirror pair [rows -> axis 0] READS them scattered and
  // loses what the fusion gains (1024^3 c128, tools/fused2_probe.py: 20.0 ms against 18.4 ms).
  int ring_probe = 0, lag_probe = 0;
  const bool pair_cols_rows = !real && !tr && !flat_out && Pu == nc && opts().fuse2 && fused2_arch_ok() &&
                              ((opts().fuse2_kinds >> FUSED_COLS_ROWS) & 1) && fused2_ring(prec, (int)n0, (int)n2, n0 * P * esz, (int)n1, &ring_probe, &lag_probe) &&
                              (prec == GFFT_F64 ? fused2_supported_f64(FUSED_COLS_ROWS, 1, (int)n0, (int)n2)
                                                : (opts().fuse2_f32 && fused2_supported_f32(FUSED_COLS_ROWS, (int)n0, (int)n2)));
  // Real transforms: forward [r2c rows -> axis 1] on the contiguous planes i0 of the flat_out schedule (FUSED_R2C_PLANES),
  // backward [axis 0 -> c2r rows] on the planes i1 (FUSED_COLS_C2R), as the complex schedule runs its last two passes
  auto real_ok = [&](int kind, int na, int nb) {
    // (real fp32 pairs: built and measured level to slightly slower than their stand-alone passes -- 1024^3 r2c f32 per step
    // 11.53 ms unfused, 11.58 with both pairs, 11.78 with the r2c pair alone, 11.42 with the c2r pair alone,
    // profiles/r04_real_pairs_f32.txt -- so they need option fuse2_f32 = 2)
    return prec == GFFT_F64 ? fused2_real_supported_f64(kind, na, nb) : (opts().fuse2_f32 >= 2 && fused2_real_supported_f32(kind, na, nb));
  };
  const bool pair_real = real && !tr && opts().fuse2 && fused2_arch_ok() && n2 % 2 == 0 &&
                         (inverse ? (((opts().fuse2_kinds >> FUSED_COLS_C2R) & 1) && real_ok(FUSED_COLS_C2R, (int)n0, (int)(n2 / 2)) &&
                                     fused2_ring(prec, (int)n0, (int)(n2 / 2), n0 * P * esz, (int)n1, &ring_probe, &lag_probe))
                                  : (((opts().fuse2_kinds >> FUSED_R2C_PLANES) & 1) && flat_out && real_ok(FUSED_R2C_PLANES, (int)(n2 / 2), (int)n1) &&
                                     fused2_ring(prec, (int)(n2 / 2), (int)n1, n1 * P * esz, (int)n0, &ring_probe, &lag_probe)));
  if (pair_real && inverse) { w_i0 = n1 * P; w_i1 = P; }     // (as under the complex pair, below)
  const bool cols_first = !inverse && pair_cols_rows;
  // ... and the workspace then is W[i0][k1][c]: the stand-alone axis-1 pass stores on NEAR strides (stores are
  // what far strides hurt), the fused pair's axis-0 tiles read the far (pitched) ones
  if (pair_cols_rows) { w_i0 = n1 * P; w_i1 = P; }
  // Tile-major workspace under that schedule (option wtile): the stand-alone axis-1 pass then WRITES every tile of 256-byte segments
  // as one contiguous 256 KiB run -- measured on the pass alone, same buffers (tools/tile_major_probe.py): 6.48 -> 5.84 ms at 1024^3
  // complex128, reading such a buffer is level -- and the pair's strided tiles read W[i0][tile][k1][.] one 256-byte row per plane.
  // In the whole schedule, same arrays (profiles/r05_ab_wtile.txt): 1024^3 c128 31.7 -> 31.0 ms per step (the pass 6.15 -> 5.85 ms,
  // the pair level), (1024,512,1024) -2.0 %, 512^3 -0.9 %, 1024^3 c64 -0.5 %; 960^3 -0.4 %, 896^3 +1.0 %, (1024,2048,1024) +0.4 %:
  // taken where axis 1 is a power of two up to 1024 (option wtile = 2: wherever the layout fits).
  const bool wtile_len = (n1 & (n1 - 1)) == 0 && n1 <= 1024;
  // (The UNFUSED complex schedules on this layout: the pass that fills W gains as much -- 768^3 c128 3.40 -> 3.15 ms, 1152^3 c64
  // 6.60 -> 5.59 -- and the in-place pass, now on the far stride, loses more -- 2.94 -> 3.51, 5.94 -> 6.35: +2.7 % / -1.4 % per step,
  // profiles/r05_ab_wtile.txt part 4; they keep W[i1][i0][c].)
  // The stand-alone forms of a voided pair read the rows tile-major: their thread layout has to advance by whole tiles.
  const bool wtile_rows = pow2_rows_nt(n2, prec) > 0 && pow2_rows_nt(n2, prec) % (int)(256 / esz) == 0 && is_pow2(n0);
  if (pair_cols_rows && opts().wtile && wtile_rows && (wtile_len || opts().wtile >= 2) && Pu % (256 / esz) == 0) {
    wtile = true;
    TWc = 256 / esz;
    pl->ws_tile = (int)TWc;
    while (((int64_t)1 << wt_lg) < TWc) ++wt_lg;
    wtS = n1 * TWc;          // (rows of padding after every tile -- 1, 3, 8 -- measured level: profiles/r05_ab_wtile.txt)
  }
  std::vector<Pass> seq;
  if (flat_out) {
    seq.push_back(rows(MODE_R2C, false, true, BUF_IN, BUF_WS));
    seq.push_back(axis1(true, true, BUF_WS, BUF_WS));
    seq.push_back(axis0_flat(BUF_WS, BUF_OUT));
  } else if (!inverse && !cols_first) {
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
    const bool half = p.d.mode == MODE_R2C_H || p.d.mode == MODE_C2R_H;
    if (half && (rc = get_twiddles(2 * (int64_t)p.d.n, prec, &p.d.rtw))) return rc;
    // (strided passes tile the line-rounded width Pu; the work model counts the nc data columns)
    const double lines = p.data_inner ? (double)p.d.batch / (double)p.d.inner * (double)p.data_inner : (double)p.d.batch;
    const double n = half ? 2.0 * p.d.n : p.d.n;          // logical length of the line
    const bool r2c = p.d.mode == MODE_R2C || p.d.mode == MODE_R2C_H, c2r = p.d.mode == MODE_C2R || p.d.mode == MODE_C2R_H;
    pl->flops += (p.d.mode == MODE_C2C ? 1.0 : 0.5) * 5.0 * n * std::log2(n) * lines;
    const double ein = r2c ? prec : esz, eout = c2r ? prec : esz;
    double nin = c2r ? (double)nc_full : n, nout = r2c ? (double)nc_full : n;
    if (p.d.tr_dir == 1) nout = p.d.tr_n;
    if (p.d.tr_dir == 2) nin = p.d.tr_n;
    pl->bytes += lines * (nin * ein + nout * eout);
    pl->passes.push_back(p);
  }
  // Complex schedules: [rows along axis 2] + [axis 0 inside the workspace] -- passes 1 + 2 forward, 2 + 3
  // backward -- as ONE persistent launch per direction, plane by plane (a plane = one i1: n0 rows of n2
  // entries), the plane handed over through the Infinity Cache instead of W (FUSED_ROWS_COLS / _COLS_ROWS).
  if (!real && !tr && !flat_out && Pu == nc && pl->passes.size() >= 3) {
    const size_t base = pl->passes.size() - 3;
    Pass &p1 = pl->passes[base], &p2 = pl->passes[base + 1], &p3 = pl->passes[base + 2];
    Pass f;
    bool ok = false;
    if (!inverse && !cols_first) {
      PassDesc dA = p1.d, dB = p2.d;                  // rows IN -> slot[i0][c];  axis 0: slot -> W[i1][k0][c]
      dA.batch = n0; dA.inner = 1; dA.in_is = 0; dA.out_is = 0; dA.out_os = P;
      dB.batch = Pu; dB.in_os = 0; dB.out_os = 0; dB.in_es = P;
      ok = make_fused2(pl, FUSED_ROWS_COLS, p1, p2, dA, dB, (int)n1, n2 * esz, w_i1 * esz, n0 * P * esz, &f);
      if (ok) { f.bytes2 = 1; pl->passes[base] = f; pl->passes.erase(pl->passes.begin() + base + 1); }
    } else {
      PassDesc dA = p2.d, dB = p3.d;                  // axis 0: W[i1][k0][c] -> slot[i0][c];  rows slot -> OUT
      dA.batch = Pu; dA.in_os = 0; dA.out_os = 0; dA.out_es = P;
      dB.batch = n0; dB.inner = 1; dB.in_is = 0; dB.out_is = 0; dB.in_os = P; dB.out_os = n1 * n2;
      int64_t a_plane = w_i1 * esz;
      if (wtile) {
        // the pair's kernels take natural-layout descriptors only (FLAGS 8192): the tile-major columns as (tile, column in tile)
        dA.mid = Pu / TWc; dA.inner = TWc;
        dA.in_ms = wtS; dA.in_is = 1; dA.in_ilg = 0; dA.in_iS = 0;
        dA.out_ms = TWc;     dA.out_is = 1; dA.out_ilg = 0; dA.out_iS = 0;
        dB.in_tlg = 0; dB.in_tS = 0;                   // (B reads the slot, natural rows)
        a_plane = TWc * esz;                           // plane k1 = one row of every tile
      }
      ok = make_fused2(pl, FUSED_COLS_ROWS, p2, p3, dA, dB, (int)n1, a_plane, n2 * esz, n0 * P * esz, &f);
      if (ok) { f.bytes2 = 1; pl->passes[base + 1] = f; pl->passes.erase(pl->passes.begin() + base + 2); }
    }
  }
  if (pair_real && pl->passes.size() >= 3) {
    const size_t base = pl->passes.size() - 3;
    // algorithmic bytes of a pass, as the loop above counts them
    auto alg = [&](const Pass &p) {
      const bool half = p.d.mode == MODE_R2C_H || p.d.mode == MODE_C2R_H;
      const double lines = p.data_inner ? (double)p.d.batch / (double)p.d.inner * (double)p.data_inner : (double)p.d.batch;
      const double n = half ? 2.0 * p.d.n : p.d.n;
      const bool r2c = p.d.mode == MODE_R2C || p.d.mode == MODE_R2C_H, c2r = p.d.mode == MODE_C2R || p.d.mode == MODE_C2R_H;
      return lines * ((c2r ? (double)nc_full : n) * (r2c ? prec : esz) + (r2c ? (double)nc_full : n) * (c2r ? prec : esz));
    };
    Pass f;
    if (!inverse) {
      Pass &p1 = pl->passes[base], &p2 = pl->passes[base + 1];
      if (p1.d.mode == MODE_R2C_H && p2.cols) {
        PassDesc dA = p1.d, dB = p2.d;                // r2c rows of plane i0: IN -> slot[i1][c];  axis 1: slot -> W[i0][k1][c]
        dA.batch = n1; dA.inner = 1; dA.in_os = n2 / 2; dA.in_is = 0; dA.out_os = P; dA.out_is = 0;
        dB.batch = Pu; dB.in_os = 0; dB.out_os = 0; dB.in_es = P;
        if (make_fused2(pl, FUSED_R2C_PLANES, p1, p2, dA, dB, (int)n0, n1 * n2 * prec, w_i0 * esz, n1 * P * esz, &f)) {
          f.bytes2 = alg(p1) + alg(p2);
          pl->passes[base] = f;
          pl->passes.erase(pl->passes.begin() + base + 1);
        }
      }
    } else {
      Pass &p2 = pl->passes[base + 1], &p3 = pl->passes[base + 2];
      if (p3.d.mode == MODE_C2R_H && p2.cols) {
        PassDesc dA = p2.d, dB = p3.d;                // axis 0 of plane i1: W -> slot[i0][c];  c2r rows: slot -> OUT
        dA.batch = Pu; dA.in_os = 0; dA.out_os = 0; dA.out_es = P;
        dB.batch = n0; dB.inner = 1; dB.in_is = 0; dB.out_is = 0; dB.in_os = P; dB.out_os = n1 * n2 / 2;
        if (make_fused2(pl, FUSED_COLS_C2R, p2, p3, dA, dB, (int)n1, w_i1 * esz, n2 * prec, n0 * P * esz, &f)) {
          f.bytes2 = alg(p2) + alg(p3);
          pl->passes[base + 1] = f;
          pl->passes.erase(pl->passes.begin() + base + 2);
        }
      }
    }
  }
  pl->fused3 = true;
  // fp32 strided passes of this schedule run between pitched rows, where two 512-thread workgroups
  // per CU on 128-byte segments (variant 2 of the n = 512 / 1024 tables: one computes while the other
  // loads) beat one 1024-thread workgroup on 256-byte segments since the 4-byte LDS bank rule --
  // whole transforms, tools/variant2_pfft_probe.py: 1024^3 r2c f32 5.78 / 5.87 -> 5.44 / 5.59 ms,
  // 512^3 c64 1.39 / 1.35 -> 1.30 / 1.30 ms, 1024^3 c64 10.7 / 10.9 -> 10.6 / 11.0 ms.  (On natural
  // power-of-two strides -- the stage arrays of multi-GPU transforms -- the wide tile stays ahead.)
  if (prec == GFFT_F32 && pl->variant_cols == 0) pl->variant_cols = 2;
  // fp64 REAL schedules keep the strided kernels of rounds 1-3 (table variant 17: 16 / 8 values per thread, two
  // exchanges): the round-4 defaults -- 32 values per thread, one exchange, non-temporal streams -- gain on complex
  // schedules (1024^3 c128 per step 32.85 -> 32.51 ms) and on natural-stride stage arrays, and lose 1.5-3 % here
  // (1024^3 r2c f64 per step 18.76 -> 19.03 / 19.31 ms, profiles/r04_real_pairs.txt)
  if (prec == GFFT_F64 && real && pl->variant_cols == 0) pl->variant_cols = 17;
  // ... and complex fp64 schedules the round-4 tiles of 16 columns (table variant 16): the 32-column tiles that win 5-12 % on natural-stride
  // arrays since round 6 (n = 256, n = 512 near strides, fft_pow2_f64.hip) LOSE inside this schedule -- its workspace is laid out in
  // 16-column tiles and pitched rows --: 512^3 c128 per step 4.159 -> 4.269 ms, 256^3 0.582 -> 0.600 ms (profiles/r06_cols_t32_probe.txt)
  if (prec == GFFT_F64 && !real && pl->variant_cols == 0) pl->variant_cols = 16;
  // XCD-contiguous tile order for the stand-alone passes of power-of-two schedules (each XCD walks its own eighth of the
  // tiles: neighbouring column chunks of the pitched workspace rows share an L2).  Neutral with the kernels of rounds 1-3
  // (+-0.2 %); with the 512- / 256-thread kernels of round 4, same arrays, plans alternating (profiles/r04_ab_swizzle.txt):
  // 1024^3 c128 per step 30.76 -> 30.52 ms and 32.23 -> 31.50 ms on two boxes (the pass 5.83 -> 5.69 ms), 1024^3 c64 19.42 ->
  // 19.00, 512^3 c128 4.17 -> 4.07, 1024^3 r2c f64 18.65 -> 18.48; level for real fp32 and 512^3 c64, +0.8 % at 768^3
  // (3^b 2^k kernels: left on automatic).  On natural-stride stage arrays (C4 on 8 GPUs) it LOSES 2-8 %: only here.
  if (pl->xcd_swizzle < 0 && is_pow2(n0) && is_pow2(n1) && is_pow2(n2) && !(real && prec == GFFT_F32)) pl->xcd_swizzle = 1;
  // Workgroups per launch.  Each walks tiles block, block + grid, ...; more, shorter walks balance the
  // tail better, too many lose the overlap of one tile's stores with the next one's loads.  Clean A/B
  // on fixed caller arrays (tools/ab_option_probe.py grid_cap ...), fwd + bwd per step: 1024^3 c128
  // 38.04 (4096) / 37.64 (16384) / 37.50 (32768) / 39.23 ms (65536 = one tile each); 512^3 c128 4.90 /
  // 4.81 / 4.79; 768^3 c128 16.96 / 16.82 / 17.26; 1024^3 c64 20.76 / 20.50 / 20.65 -- but real
  // transforms the other way (1024^3 r2c f64 20.60 / 20.70 / 21.21, f32 11.23 / 11.79): complex
  // schedules take 16384, real ones 8192 (1024^3 r2c f64 20.60 -> 20.48, f32 11.09 -> 11.04 ms; 2048 and
  // 1024 lose 1-2 % each), stand-alone strided passes keep 4096 (GFFT_GRID_CAP overrides them all).
  if (opts().grid_cap <= 0)
    for (Pass &p : pl->passes) p.d.grid_cap = real ? 8192 : ((is_pow2(n0) && is_pow2(n1) && is_pow2(n2)) ? 32768 : 16384);     // (R4, XCD-contiguous order, 32-value kernels: 1024^3 c128 per step 32.23 (4096) / 31.97 (16384) / 31.81 (32768) / 32.07 ms (65536))
  return GFFT_OK;
}

hipError_t run_pass(const gfft_plan_s *pl, const Pass &p, const PassDesc &d0, const void *in, void *out,
                    double scale, hipStream_t s) {
  if (p.kind == PK_EMBED) return launch_embed(p.pt, pl->precision, in, out, s);
  if (p.kind == PK_MULB) return launch_mulb(p.pt, pl->precision, out, s);
  if (p.kind == PK_EXTRACT) return launch_extract(p.pt, pl->precision, in, out, scale * p.extra_scale, s);
  PassDesc d = d0;
  // auto: only where a strided pass WRITES rows that do not start on 128-byte lines (odd-width
  // half spectra): neighbouring chunks then meet in one L2 and their partial lines merge.  For a
  // pass that only READS such rows the order lowers FETCH_SIZE (1.44 -> 1.13 x the algorithmic
  // bytes) but not the time (3.48 ms plain, 3.65 ms XCD-contiguous, tools/flat_probe.py): off.
  // (measured 1024^3 r2c: 5.4 -> 5.0 ms fp64, 3.6 -> 2.6 ms fp32; neutral-to-slightly-negative on
  // aligned arrays, so it stays off there)
  const int64_t esz_out = ((d.mode == MODE_C2R || d.mode == MODE_R2R) ? 1 : 2) * (int64_t)pl->precision;
  const int64_t esz_in = ((d.mode == MODE_R2C || d.mode == MODE_R2R) ? 1 : 2) * (int64_t)pl->precision;
  (void)esz_in;
  // workgroups per launch of stand-alone row passes (the first / last stage of every multi-GPU
  // transform): tools/ab_gridcap_serial.py, caps alternated on one plan and the same arrays --
  // (256,512,1024) c128 rows 0.803 (4096) -> 0.782 ms (16384), (512,1024,2048) f32 r2c rows 1.854 ->
  // 1.803 ms; strided stand-alone passes are level or lose (far axis 1.001 -> 1.037 ms) and keep 4096
  if (!d.grid_cap && p.regk && !p.cols && !pl->fused3 && opts().grid_cap <= 0) d.grid_cap = 16384;
  // ... and of strided passes that READ on a far power-of-two stride (the caller's natural array along its outermost axis: the far
  // stage of every backward distributed transform): more, shorter walks -- (1024,256,512) complex128 natural -> natural 0.895 ->
  // 0.849 ms, natural -> pitched 0.903 -> 0.857 ms; pitched inputs and near axes are level or lose (tools/bwd_probe.py,
  // profiles/r06_bwd_probe.txt)
  if (!d.grid_cap && p.regk && p.cols && !pl->fused3 && opts().grid_cap <= 0 && d.mid == 1 && !d.flat && !d.in_lgp &&
      d.in_es >= ((int64_t)1 << 16) && (d.in_es & (d.in_es - 1)) == 0) d.grid_cap = 16384;
  d.swizzle = pl->xcd_swizzle >= 0 ? pl->xcd_swizzle : (p.cols && !d.flat && ((d.out_es * esz_out) % 128 != 0) ? 1 : 0);
  if ((d.mode == MODE_R2C_H || d.mode == MODE_C2R_H) && real_half_supported(d.n))
    return pl->precision == 8 ? launch_real_half_f64(d, pl->variant_rows, in, out, s)
                              : launch_real_half_f32(d, pl->variant_rows, in, out, s);
  if ((d.mode == MODE_R2C_H || d.mode == MODE_C2R_H) && mixv_supported(d.n))
    return pl->precision == 8 ? launch_real_half_mixv_f64(d, in, out, s) : launch_real_half_mixv_f32(d, in, out, s);
  if (d.mode == MODE_R2C_H || d.mode == MODE_C2R_H)
    return pl->precision == 8 ? launch_real_half_mix_f64(d, in, out, s) : launch_real_half_mix_f32(d, in, out, s);
  if (p.regk && d.mode == MODE_R2R && pow2_r2r_supported(d.n))
    return pl->precision == 8 ? launch_pow2_r2r_f64(d, p.cols, in, out, s) : launch_pow2_r2r_f32(d, p.cols, in, out, s);
  if (p.regk && mix3_supported(d.n)) {
    return pl->precision == 8 ? launch_mix3_f64(d, p.cols, in, out, s) : launch_mix3_f32(d, p.cols, in, out, s);
  }
  if (p.regk && mix5_supported(d.n)) {
    return pl->precision == 8 ? launch_mix5_f64(d, p.cols, in, out, s) : launch_mix5_f32(d, p.cols, in, out, s);
  }
  if (p.regk && mixv_supported(d.n)) {
    return pl->precision == 8 ? launch_mixv_f64(d, p.cols, pl->mixv_variant, in, out, s) : launch_mixv_f32(d, p.cols, pl->mixv_variant, in, out, s);
  }
  if (p.regk) {
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

// Batched 2-D complex transform plane by plane: the passes of gfft_plan_create_guru2 (see there), pushed onto pl->passes
// -- one fused launch where a pair exists, else (unless fused_only) the two stand-alone passes [first: IN -> OUT, second:
// in place on OUT].
static int build_pair2d(gfft_plan_s *pl, const gfft_iodim *cols, int64_t n2, const gfft_iodim *planes, bool inverse, int cols_first,
                        int in_blocks, int64_t in_block_stride, int out_blocks, int64_t out_block_stride, bool fused_only, bool *fused_out) {
  const int precision = pl->precision;
  const int64_t esz = 2 * (int64_t)precision;
  const int64_t n1 = cols->n, np = planes->n;
  auto lg2 = [](int v) { int l = 0; while ((1 << l) < v) ++l; return l; };
  // one side of the array pair as the two passes see it: the strided pass steps es along the axis and jumps between
  // blocks; the row pass places row r = (block m, row i of the block) by its batch strides
  struct Side { int nb; int64_t per, es, bstride, plane; };
  const Side si{in_blocks, n1 / in_blocks, cols->is, in_blocks > 1 ? in_block_stride : (n1 / in_blocks) * cols->is, planes->is};
  const Side so{out_blocks, n1 / out_blocks, cols->os, out_blocks > 1 ? out_block_stride : (n1 / out_blocks) * cols->os, planes->os};

  auto base = [&](int64_t n, bool strided) {
    Pass p;
    p.regk = true;
    p.cols = strided;
    p.logical_first = true;
    p.d.n = (int)n;
    p.d.mode = MODE_C2C;
    p.d.conj_in = p.d.conj_out = inverse ? 1 : 0;
    p.d.mid = 1;
    p.d.inner = 1;
    p.d.in_is = p.d.out_is = 1;
    p.d.in_es = p.d.out_es = 1;
    p.d.scale = 1.0;
    return p;
  };
  auto cols_in = [&](Pass &p, const Side &s) {
    p.d.in_os = s.plane;  p.d.in_es = s.es;
    if (s.nb > 1) { p.d.in_lgp = lg2(s.nb);  p.d.in_jump = s.bstride - s.per * s.es; }
    p.blocks[0] = s.nb;  p.bstride[0] = s.nb > 1 ? s.bstride : 0;
  };
  auto cols_out = [&](Pass &p, const Side &s) {
    p.d.out_os = s.plane;  p.d.out_es = s.es;
    if (s.nb > 1) { p.d.out_lgp = lg2(s.nb);  p.d.out_jump = s.bstride - s.per * s.es; }
    p.blocks[1] = s.nb;  p.bstride[1] = s.nb > 1 ? s.bstride : 0;
  };
  // the stand-alone forms: the first pass carries the data across (IN -> OUT), the second runs in place on OUT
  Pass pr = base(n2, false), pc = base(n1, true);
  pc.d.batch = np * n2;
  pc.d.inner = n2;
  pr.d.batch = np * n1;
  const Side &rs = (si.nb > 1 && !cols_first) ? si : so;        // the blocks the row pass walks its rows by
  pr.d.mid = rs.nb;  pr.d.inner = rs.per;
  if (!cols_first) {
    pr.d.in_os = si.plane;   pr.d.in_ms = si.nb > 1 ? si.bstride : rs.per * si.es;  pr.d.in_is = si.es;
    pr.d.out_os = so.plane;  pr.d.out_ms = so.nb > 1 ? so.bstride : rs.per * so.es;  pr.d.out_is = so.es;
    pr.src = BUF_IN;  pr.dst = BUF_OUT;
    cols_in(pc, so);  cols_out(pc, so);
    pc.src = BUF_OUT;  pc.dst = BUF_OUT;
  } else {
    cols_in(pc, si);  cols_out(pc, so);
    pc.src = BUF_IN;  pc.dst = BUF_OUT;
    pr.d.in_os = pr.d.out_os = so.plane;
    pr.d.in_ms = pr.d.out_ms = so.bstride;
    pr.d.in_is = pr.d.out_is = so.es;
    pr.src = BUF_OUT;  pr.dst = BUF_OUT;
  }
  int rc = get_twiddles(n2, precision, &pr.d.tw);
  if (!rc) rc = get_twiddles(n1, precision, &pc.d.tw);
  if (rc) return rc;
  // ... and as ONE persistent launch, plane by plane through the Infinity Cache: slot[row of the plane][P]
  bool fused = false;
  if (np < ((int64_t)1 << 30)) {
    int64_t P = n2;                                       // slot rows pitched off the power of two
    if ((P * esz) % 2048 == 0) P += 256 / esz;
    PassDesc dA, dB;
    Pass f;
    if (!cols_first) {
      dA = pr.d;  dB = pc.d;
      dA.batch = n1;  dA.in_os = 0;  dA.out_os = 0;  dA.out_ms = rs.per * P;  dA.out_is = P;
      dB.batch = n2;  dB.in_os = 0;  dB.out_os = 0;  dB.in_es = P;  dB.in_lgp = 0;  dB.in_jump = 0;
      fused = make_fused2(pl, so.nb > 1 ? FUSED_PLANES_2D_B : FUSED_PLANES_2D, pr, pc, dA, dB, (int)np, si.plane * esz, so.plane * esz, n1 * P * esz, &f);
    } else {
      dA = pc.d;  dB = pr.d;
      dA.batch = n2;  dA.in_os = 0;  dA.out_os = 0;  dA.out_es = P;  dA.out_lgp = 0;  dA.out_jump = 0;
      dB.batch = n1;  dB.in_os = 0;  dB.out_os = 0;  dB.in_ms = rs.per * P;  dB.in_is = P;
      fused = make_fused2(pl, si.nb > 1 ? FUSED_PLANES_CR_B : FUSED_COLS_ROWS, pc, pr, dA, dB, (int)np, si.plane * esz, so.plane * esz, n1 * P * esz, &f);
    }
    if (fused) pl->passes.push_back(f);
  }
  if (!fused && !fused_only) {
    if (!cols_first) { pl->passes.push_back(pr); pl->passes.push_back(pc); }
    else { pl->passes.push_back(pc); pl->passes.push_back(pr); }
  }
  *fused_out = fused;
  return GFFT_OK;
}

extern "C" {

const char *gfft_strerror(int status) {
  switch (status) {
    case GFFT_OK: return "success";
    case GFFT_ERR_INVALID: return "invalid argument";
    case GFFT_ERR_UNSUPPORTED: return "unsupported transform";
    case GFFT_ERR_NO_DEVICE: return "no HIP device";
    case GFFT_ERR_HIP: return "HIP runtime error";
    case GFFT_ERR_NOMEM: return "out of memory";
    case GFFT_ERR_VOIDED: return "asynchronous launch failure";
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
  else if (!strcmp(key, "plane2d")) opts().plane2d = value;
  else if (!strcmp(key, "fuse2")) opts().fuse2 = value;
  else if (!strcmp(key, "fuse2_ring")) opts().fuse2_ring = value;
  else if (!strcmp(key, "fuse2_lag")) opts().fuse2_lag = value;
  else if (!strcmp(key, "fuse2_kinds")) opts().fuse2_kinds = value;
  else if (!strcmp(key, "fuse2_wait_ms")) opts().fuse2_wait_ms = value;
  else if (!strcmp(key, "fuse2_f32")) opts().fuse2_f32 = value;
  else if (!strcmp(key, "fuse2_n512")) gfft::g_fuse2_n512 = value;
  else if (!strcmp(key, "c2r_2048")) gfft::g_c2r_2048 = value;
  else if (!strcmp(key, "fuse2_mixv")) gfft::g_fuse2_mixv = value;
  else if (!strcmp(key, "fuse2_mixed")) gfft::g_fuse2_mixed = value;
  else if (!strcmp(key, "fuse2_f32_n512")) gfft::g_fuse2_f32_n512 = value;
  else if (!strcmp(key, "debug_flat")) opts().debug_flat = value;
  else if (!strcmp(key, "flat_out")) opts().flat_out = value;
  else if (!strcmp(key, "wtile")) opts().wtile = value;
  else if (!strcmp(key, "mixv")) opts().mixv = value;
  else if (!strcmp(key, "mixv_variant")) opts().mixv_variant = value;
  else if (!strcmp(key, "debug_tile_lg")) opts().debug_tile_lg = value;
  else if (!strcmp(key, "debug_tile_side")) opts().debug_tile_side = value;
  else if (!strcmp(key, "debug_tile_stride")) opts().debug_tile_stride = value;
  else if (!strcmp(key, "profile")) opts().profile = value;
  else if (!strcmp(key, "debug_tw_index")) opts().debug_tw_index = value;
  else if (!strcmp(key, "debug_tw_exp")) opts().debug_tw_exp = value;
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
  pl->mixv_variant = opts().mixv_variant;
  pl->xcd_swizzle = opts().xcd_swizzle;

  rc = GFFT_OK;
  auto line = [&](int axis, int mode, bool inverse, const std::vector<int64_t> &sin_,
                  const std::vector<int64_t> &sout_, int src, int dst) {
    Line L;
    L.outer = L.inner = 1;
    for (int i = 0; i < axis; ++i) L.outer *= sin_[i];
    for (int i = axis + 1; i < ndims; ++i) L.inner *= sin_[i];
    L.nin = sin_[axis];
    L.nout = sout_[axis];
    L.n = (mode == MODE_C2R) ? sout_[axis] : sin_[axis];
    L.mode = mode;
    L.inverse = inverse;
    L.src = src;
    L.dst = dst;
    return plan_line(pl, L, true);
  };
  if (fused3_applicable(pl)) {
    rc = plan_fused3(pl);
  } else if (kind == GFFT_C2C_FORWARD || kind == GFFT_C2C_BACKWARD) {
    const bool inv = kind == GFFT_C2C_BACKWARD;
    for (int i = naxes - 1; i >= 0 && !rc; --i)
      rc = line(ax[i], MODE_C2C, inv, pl->sizes_in, pl->sizes_in, i == naxes - 1 ? BUF_IN : BUF_OUT, BUF_OUT);
    // Batched 2-D transforms over the last two axes of a 3-D array -- the leading stage of a slab-decomposed
    // PFFT with collapse=True, or fftn(axes=(1, 2)) --: plane by plane in one fused launch, [strided along axis 1 ->
    // rows along axis 2] in both directions (build_pair2d: strided reads, whole rows written -- round 6; against the
    // [rows -> strided] order of rounds 4-5: level in complex128, 9 % ahead in complex64, profiles/r06_stage_probe_slab.txt)
    if (!rc && ndims == 3 && naxes == 2 && ax[0] == 1 && ax[1] == 2 && pl->passes.size() == 2) {
      const Pass &pr = pl->passes[0], &pc = pl->passes[1];
      const int64_t n0 = sizes_in[0], n1 = sizes_in[1], n2 = sizes_in[2];
      if (pr.kind == PK_FFT && pc.kind == PK_FFT && pr.regk && pc.regk && !pr.cols && pc.cols && !pr.d.tw_hi && !pc.d.tw_hi &&
          is_pow2(n1) && is_pow2(n2) && n0 < ((int64_t)1 << 30)) {
        const std::vector<Pass> two = pl->passes;
        pl->passes.clear();
        const gfft_iodim c{n1, n2, n2}, p{n0, n1 * n2, n1 * n2};
        bool fused = false;
        if (build_pair2d(pl, &c, n2, &p, inv, 1, 1, 0, 1, 0, true, &fused) != GFFT_OK || !fused) pl->passes = two;
      }
    }
    // Planes of 32 x 32 / 64 x 64 points -- config C1's 64^3, small images --: rows and columns of a plane in ONE launch with the
    // plane held in LDS (fft_plane2d.hip): such arrays live in the caches and a pass costs its dependent memory round trip, not
    // its bytes (64^3 complex128: three passes of ~5 us each, tools/c1_probe.py); this removes one of them.
    // (a 2-D array is one plane)
    const int p0 = ndims - 2;                                // the two in-plane axes: p0, p0 + 1
    if (!rc && (ndims == 3 || ndims == 2) && naxes >= 2 && ax[naxes - 2] == p0 && ax[naxes - 1] == p0 + 1 &&
        (naxes == 2 || (ndims == 3 && naxes == 3 && ax[0] == 0)) &&
        sizes_in[p0] == sizes_in[p0 + 1] && plane2d_supported((int)sizes_in[p0], precision) && pl->passes.size() == (size_t)naxes && opts().plane2d) {
      const Pass &pr = pl->passes[0], &pc = pl->passes[1];
      const int64_t nplanes = ndims == 3 ? sizes_in[0] : 1;
      if (pr.kind == PK_FFT && pc.kind == PK_FFT && pr.regk && pc.regk && !pr.cols && pc.cols && nplanes < ((int64_t)1 << 30)) {
        const int64_t n = sizes_in[p0], P = plane2d_pitch((int)n), esz = 2 * (int64_t)precision;
        Pass f = pr;
        f.kind = PK_PLANE2D;
        f.d = pr.d;                                          // rows of ONE plane: natural rows -> the plane in LDS, rows P entries apart
        f.d.batch = n;  f.d.mid = 1;  f.d.inner = 1;  f.d.in_os = n;  f.d.in_is = 0;  f.d.out_os = P;  f.d.out_is = 0;
        f.d2 = pc.d;                                         // columns of one plane: LDS -> natural rows
        f.d2.batch = n;  f.d2.mid = 1;  f.d2.inner = n;  f.d2.in_os = 0;  f.d2.out_os = 0;  f.d2.in_es = P;  f.d2.in_is = 1;
        f.fused.planes = (int)nplanes;
        f.fused.a_in_plane = n * n * esz;
        f.fused.b_out_plane = n * n * esz;
        f.src = BUF_IN;  f.dst = BUF_OUT;
        f.bytes2 = 4.0 * (double)nplanes * (double)n * (double)n * (double)esz;
        pl->passes.erase(pl->passes.begin(), pl->passes.begin() + 2);
        pl->passes.insert(pl->passes.begin(), f);
      }
    }
  } else if (kind == GFFT_R2C) {
    rc = line(last, MODE_R2C, false, pl->sizes_in, pl->sizes_out, BUF_IN, BUF_OUT);
    for (int i = naxes - 2; i >= 0 && !rc; --i)
      rc = line(ax[i], MODE_C2C, false, pl->sizes_out, pl->sizes_out, BUF_OUT, BUF_OUT);
  } else {
    // C2R: the complex passes run into / inside the WS region, so the caller's input is never
    // written (FFTW's multi-dimensional c2r destroys it), then the real pass
    for (int i = 0; i <= naxes - 2 && !rc; ++i)
      rc = line(ax[i], MODE_C2C, true, pl->sizes_in, pl->sizes_in, i == 0 ? BUF_IN : BUF_WS, BUF_WS);
    if (!rc) rc = line(last, MODE_C2R, true, pl->sizes_in, pl->sizes_out, naxes > 1 ? BUF_WS : BUF_IN, BUF_OUT);
    if (naxes > 1) {
      size_t bytes = 2 * (size_t)precision;
      for (int i = 0; i < ndims; ++i) bytes *= (size_t)sizes_in[i];
      need(pl, BUF_WS, bytes);
    }
  }
  if (rc) {
    delete pl;
    return rc;
  }
  // the scale factor rides on the last pass
  pl->passes.back().carries_scale = true;
  *plan = pl;
  ++g_live_plans;
  return GFFT_OK;
}

int gfft_plan_create_padded(gfft_plan *plan, const int64_t *padded, const int64_t *kept, int kind, int precision) {
  if (!plan || !padded || !kept) return fail(GFFT_ERR_INVALID, "null argument");
  *plan = nullptr;
  if (precision != GFFT_F32 && precision != GFFT_F64) return fail(GFFT_ERR_INVALID, "precision must be 4 or 8");
  if (kind != GFFT_C2C_FORWARD && kind != GFFT_C2C_BACKWARD && kind != GFFT_R2C && kind != GFFT_C2R)
    return fail(GFFT_ERR_INVALID, "kind must be c2c forward / backward, r2c or c2r");
  const bool real = kind == GFFT_R2C || kind == GFFT_C2R;
  const bool inverse = kind == GFFT_C2C_BACKWARD || kind == GFFT_C2R;
  for (int i = 0; i < 3; ++i) {
    const int64_t of = (real && i == 2) ? padded[i] / 2 + 1 : padded[i];
    if (padded[i] < 1 || kept[i] < 1 || kept[i] > of) return fail(GFFT_ERR_INVALID, "kept entries must lie in 1 .. transformed entries");
  }
  int rc = check_device();
  if (rc) return rc;
  // the all-axes schedule needs one register-kernel pass per axis (packed-real rows on a real axis)
  if (!opts().fused3) return fail(GFFT_ERR_UNSUPPORTED, "fused 3-D plans are switched off");
  for (int i = 0; i < 3; ++i) {
    const bool real_axis = i == 2 && real;
    const bool one_pass = regk_ok(padded[i], precision) ||
                          (real_axis ? (padded[i] % 2 == 0 && real_half_mixv_ok(padded[i] / 2)) : regk_c2c_ok(padded[i], precision, MODE_C2C));
    if (!one_pass || (kept[i] < (real_axis ? padded[i] / 2 + 1 : padded[i]) && !fused_pad_ok(real_axis ? padded[i] / 2 : padded[i])))
      return fail(GFFT_ERR_UNSUPPORTED, "padded length without a single-pass kernel");
  }
  if (real && !(padded[2] % 2 == 0 &&
                (real_half_supported((int)(padded[2] / 2)) || real_half_mix_supported((int)(padded[2] / 2)) || real_half_mixv_ok(padded[2] / 2))))
    return fail(GFFT_ERR_UNSUPPORTED, "real axis without a packed-real row kernel");
  const int64_t bytes = padded[0] * padded[1] * padded[2] * (real ? 1 : 2) * precision;
  if (bytes < opts().fused3_min_bytes) return fail(GFFT_ERR_UNSUPPORTED, "below the size where the workspace schedule pays");
  gfft_plan_s *pl = new gfft_plan_s;
  pl->ndims = 3;
  pl->kind = kind;
  pl->precision = precision;
  std::vector<int64_t> phys(padded, padded + 3), spec(kept, kept + 3);
  pl->sizes_in = inverse ? spec : phys;
  pl->sizes_out = inverse ? phys : spec;
  pl->axes = {0, 1, 2};
  pl->trunc = spec;
  pl->variant_rows = opts().variant_rows;
  pl->variant_cols = opts().variant_cols;
  pl->mixv_variant = opts().mixv_variant;
  pl->xcd_swizzle = opts().xcd_swizzle;
  rc = plan_fused3(pl);
  if (rc) {
    delete pl;
    return rc;
  }
  pl->passes.back().carries_scale = true;
  *plan = pl;
  ++g_live_plans;
  return GFFT_OK;
}

int gfft_plan_create_r2r(gfft_plan *plan, int ndims, const int64_t *sizes, int naxes, const int *axes,
                         const int *kinds, int precision) {
  if (!plan || !sizes || !axes || !kinds) return fail(GFFT_ERR_INVALID, "null argument");
  *plan = nullptr;
  if (ndims < 1 || ndims > 16 || naxes < 1 || naxes > ndims) return fail(GFFT_ERR_INVALID, "bad ndims/naxes");
  if (precision != GFFT_F32 && precision != GFFT_F64) return fail(GFFT_ERR_INVALID, "precision must be 4 or 8");
  std::vector<int> ax(axes, axes + naxes);
  std::vector<char> seen(ndims, 0);
  for (int i = 0; i < naxes; ++i) {
    int &a = ax[i];
    if (a < 0) a += ndims;
    if (a < 0 || a >= ndims || seen[a]) return fail(GFFT_ERR_INVALID, "bad or repeated axis");
    seen[a] = 1;
    if (kinds[i] < GFFT_REDFT00 || kinds[i] > GFFT_RODFT11)
      return fail(GFFT_ERR_UNSUPPORTED, "r2r kind must be one of REDFT00..RODFT11 (halfcomplex / DHT kinds are not implemented)");
  }
  for (int i = 0; i < ndims; ++i)
    if (sizes[i] < 1) return fail(GFFT_ERR_INVALID, "sizes must be >= 1");
  int rc = check_device();
  if (rc) return rc;
  gfft_plan_s *pl = new gfft_plan_s;
  pl->ndims = ndims;
  pl->kind = GFFT_R2R;
  pl->precision = precision;
  pl->sizes_in.assign(sizes, sizes + ndims);
  pl->sizes_out = pl->sizes_in;
  pl->axes = ax;
  pl->variant_rows = opts().variant_rows;
  pl->variant_cols = opts().variant_cols;
  pl->mixv_variant = opts().mixv_variant;
  pl->xcd_swizzle = opts().xcd_swizzle;
  for (int i = naxes - 1; i >= 0 && !rc; --i) {
    int64_t outer = 1, inner = 1;
    for (int k = 0; k < ax[i]; ++k) outer *= sizes[k];
    for (int k = ax[i] + 1; k < ndims; ++k) inner *= sizes[k];
    rc = plan_r2r_line(pl, outer, sizes[ax[i]], inner, kinds[i], i == naxes - 1 ? BUF_IN : BUF_OUT, BUF_OUT);
  }
  if (rc) {
    delete pl;
    return rc;
  }
  pl->passes.back().carries_scale = true;
  *plan = pl;
  ++g_live_plans;
  return GFFT_OK;
}

int gfft_execute(gfft_plan pl, const void *d_in, void *d_out, double scale, void *stream) {
  if (!pl || !d_in || !d_out) return fail(GFFT_ERR_INVALID, "null argument");
  {
    // a fused launch of an earlier execution of THIS plan gave up a wait: report it now, once, and enqueue nothing (the
    // plan has been switched to stand-alone passes; calling again runs it that way).  Another plan's pending event stays
    // with that plan (gfft_plan_status / gfft_async_error / its own next execute) and does not refuse this call.
    int rc = poll_async_error(pl);
    if (rc) return rc;
  }
  if (current_device() != pl->device) return fail(GFFT_ERR_INVALID, "the plan was made on another device than the calling thread's current one");
  if ((pl->kind == GFFT_R2C || pl->kind == GFFT_C2R) && d_in == d_out)
    return fail(GFFT_ERR_INVALID, "in-place real transforms are not supported");
  if (!pl->fused_off && pl->async_slot < 0) {
    bool any = false;
    for (const Pass &q : pl->passes) any = any || q.kind == PK_FUSED2;
    if (any) {
      // this plan's own word of the host-visible table: two live plans never share one (ids 64 apart used to), so a
      // voided launch is always attributed; with every word taken the pairs run as their stand-alone passes instead
      AsyncErrors &ae = async_errors();
      std::lock_guard<std::mutex> lock(ae.m);
      if (ae.word())
        for (int i = 0; i < AsyncErrors::SLOTS && pl->async_slot < 0; ++i)
          if (!ae.owner[i]) { ae.owner[i] = pl; pl->async_slot = i; }
      if (pl->async_slot < 0) {
        pl->fused_off = true;
        for (const Pass &q : pl->passes)
          if (q.kind == PK_FUSED2 && q.alt_buf >= 0 && q.alt_bytes > pl->region_bytes[q.alt_buf]) pl->region_bytes[q.alt_buf] = q.alt_bytes;
      }
    }
  }
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  // scratch regions: carved from the stream's shared buffer (scratch_pool)
  size_t off[BUF_COUNT] = {0, 0, 0, 0, 0, 0, 0}, total = 0;
  for (int b = BUF_WS; b < BUF_COUNT; ++b) {
    off[b] = total;
    total += align256(pl->region_bytes[b]);
  }
  void *scratch = nullptr;
  if (total) {
    int rc = scratch_pool().get(s, total, &scratch);
    if (rc) return rc;
  }
  void *bufs[BUF_COUNT] = {const_cast<void *>(d_in), d_out, nullptr, nullptr, nullptr, nullptr, nullptr};
  for (int b = BUF_WS; b < BUF_COUNT; ++b)
    if (pl->region_bytes[b]) bufs[b] = static_cast<char *>(scratch) + off[b];

  std::vector<hipEvent_t> *ev = nullptr;
  auto mark = [&]() -> hipError_t {
    if (!ev) return hipSuccess;
    hipEvent_t e;
    hipError_t rc = hipEventCreate(&e);
    if (rc != hipSuccess) return rc;
    rc = hipEventRecord(e, s);
    ev->push_back(e);
    return rc;
  };
  if (opts().profile) {
    pl->prof.emplace_back();
    ev = &pl->prof.back();
    HIP_TRY(mark());
  }
  for (const Pass &p : pl->passes) {
    PassDesc d = p.d;
    if (p.kind == PK_PLANE2D) {
      PassDesc d2 = p.d2;
      d2.scale = p.carries_scale ? scale : 1.0;
      HIP_TRY(launch_plane2d(d, d2, pl->precision, p.fused.planes, p.fused.a_in_plane, p.fused.b_out_plane, bufs[p.src], bufs[p.dst], s));
      HIP_TRY(mark());
      continue;
    }
    if (p.kind == PK_FUSED2 && pl->fused_off) {
      // the pair as its two stand-alone passes (a fused launch of this plan gave up a wait, poll_async_error)
      for (int k = 0; k < 2; ++k) {
        const Pass &q = pl->alt[p.alt_first + k];
        PassDesc dq = q.d;
        const bool sc = k == 1 && p.carries_scale;
        if (sc) dq.scale = scale;
        HIP_TRY(run_pass(pl, q, dq, bufs[q.src], bufs[q.dst], sc ? scale : 1.0, s));
      }
      HIP_TRY(mark());
      continue;
    }
    if (p.kind == PK_FUSED2) {
      PassDesc d2 = p.d2;
      if (p.carries_scale) d2.scale = scale;
      FusedDesc f = p.fused;
      char *ring = static_cast<char *>(bufs[BUF_RING]);
      f.ctr = reinterpret_cast<unsigned *>(ring + (size_t)f.ring * (size_t)f.slot_bytes);
      {
        AsyncErrors &ae = async_errors();
        f.host_flag = (ae.flag && pl->async_slot >= 0) ? ae.flag + pl->async_slot : nullptr;
        const int ms = opts().fuse2_wait_ms;
        f.wait_ticks = ms <= 0 ? 0u : (ms > 40000 ? 4000000000u : (unsigned)ms * 100000u);      // 100 MHz ticks
      }
      static const int debug = getenv("GFFT_FUSE2_DEBUG") ? atoi(getenv("GFFT_FUSE2_DEBUG")) : 0;
      if (debug) { f.wait_ticks = 100000u; f.host_flag = nullptr; f.debug = (unsigned)debug; }      // (1 ms; counters printed below)
      if (pl->precision == GFFT_F32 && (p.fused_kind == FUSED_R2C_PLANES || p.fused_kind == FUSED_COLS_C2R))
        HIP_TRY(launch_fused2_real_f32(p.fused_kind, d, d2, p.dev_descs, f, bufs[p.src], ring, bufs[p.dst], s));
      else if (pl->precision == GFFT_F32)
        HIP_TRY(launch_fused2_f32(p.fused_kind, d, d2, p.dev_descs, f, bufs[p.src], ring, bufs[p.dst], s));
      else if (p.fused_kind == FUSED_R2C_PLANES || p.fused_kind == FUSED_COLS_C2R)
        HIP_TRY(launch_fused2_real_f64(p.fused_kind, d, d2, p.dev_descs, f, bufs[p.src], ring, bufs[p.dst], s));
      else
        HIP_TRY(launch_fused2_f64(p.fused_kind, p.fused_variant, d, d2, p.dev_descs, f, bufs[p.src], ring, bufs[p.dst], s));
      if (debug) {      // developer aid: the launch's counters (tickets drawn, waits given up, tiles per plane)
        HIP_TRY(hipStreamSynchronize(s));
        std::vector<unsigned> h(16 + 2 * (size_t)f.planes);
        HIP_TRY(hipMemcpy(h.data(), f.ctr, h.size() * sizeof(unsigned), hipMemcpyDeviceToHost));
        unsigned long long sa = 0, sb = 0;
        for (int i = 0; i < f.planes; ++i) { sa += h[16 + i]; sb += h[16 + f.planes + i]; }
        fprintf(stderr, "[gfft fuse2] kind %d planes %d tiles %d+%d ring %d lag %d: tickets %u, waits given up %u, A tiles %llu, B tiles %llu\n",
                p.fused_kind, f.planes, f.tiles_a, f.tiles_b, f.ring, f.lag, h[0], h[1], sa, sb);
#ifdef GFFT_FUSE2_TRACE
        if (const char *path = getenv("GFFT_FUSE2_TRACE_FILE")) {
          std::vector<unsigned long long> tr((size_t)1024 * 96 * 16);
          const size_t off = ((16 + 2 * (size_t)f.planes + 63) & ~(size_t)63) * sizeof(unsigned);
          HIP_TRY(hipMemcpy(tr.data(), reinterpret_cast<char *>(f.ctr) + off, tr.size() * sizeof(tr[0]), hipMemcpyDeviceToHost));
          if (FILE *fp = fopen(path, "wb")) { fwrite(tr.data(), sizeof(tr[0]), tr.size(), fp); fclose(fp); }
        }
#endif
      }
      HIP_TRY(mark());
      continue;
    }
    if (p.carries_scale) d.scale = scale;
    HIP_TRY(run_pass(pl, p, d, bufs[p.src], bufs[p.dst], p.carries_scale ? scale : 1.0, s));
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
  static const char *kinds[] = {"", "embed", "mul-B", "extract"};
  if (p.kind == PK_FUSED2) {
    // two axis passes in one launch: the algorithmic bytes of both (one read + one write of the array each)
    static const char *fk[] = {"fused rows+cols", "fused cols+rows", "fused four-step", "fused 2-D rows+cols", "fused four-step (rows)",
                               "fused r2c-rows+cols", "fused cols+c2r-rows", "fused 2-D rows+cols(blocks)", "fused 2-D cols(blocks)+rows"};
    snprintf(buf, len, "%s n=%dx%d", fk[p.fused_kind], p.d.n, p.d2.n);
    if (bytes) *bytes = p.bytes2 > 1 ? p.bytes2 : (double)p.fused.planes * 2.0 * pl->precision *
                        ((double)p.d.batch * 2.0 * p.d.n + (double)p.d2.batch * 2.0 * p.d2.n);
    return GFFT_OK;
  }
  if (p.kind == PK_PLANE2D) {
    snprintf(buf, len, "planes in LDS rows+cols n=%dx%d", p.d.n, p.d2.n);
    if (bytes) *bytes = p.bytes2;
    return GFFT_OK;
  }
  if (p.kind != PK_FFT) {
    snprintf(buf, len, "%s n=%lld", kinds[p.kind], (long long)p.pt.n);
    if (bytes) *bytes = 0;
    return GFFT_OK;
  }
  const bool percol = !p.regk && (p.d.n == 3 || p.d.n == 5 || p.d.n == 7 || p.d.n == 11 || p.d.n == 13) &&
                      p.d.mode == MODE_C2C && !p.d.tw_hi && p.d.in_is == 1 && p.d.out_is == 1 && p.d.inner >= 64;
  snprintf(buf, len, "%s n=%d", p.regk ? (p.cols ? "pow2-cols" : "pow2-rows") : (percol ? "percol" : "generic"), p.d.n);
  if (bytes && (p.d.mode == MODE_R2C_H || p.d.mode == MODE_C2R_H)) {
    // n complex = 2n reals on one side, n + 1 complex on the other
    *bytes = (double)p.d.batch * (2.0 * p.d.n * pl->precision + (p.d.n + 1.0) * 2.0 * pl->precision);
    snprintf(buf, len, "real-rows n=%d", 2 * p.d.n);
    return GFFT_OK;
  }
  if (bytes) {
    const double esz = 2.0 * pl->precision;
    const double nc = p.d.mode == MODE_C2C ? p.d.n : p.d.n / 2 + 1;
    const double ein = p.d.mode == MODE_R2C ? p.d.n * (double)pl->precision : nc * esz;
    const double eout = p.d.mode == MODE_C2R ? p.d.n * (double)pl->precision : nc * esz;
    *bytes = (double)p.d.batch * (ein + eout);
    if (p.data_inner) *bytes *= (double)p.data_inner / (double)p.d.inner;
    if (p.d.mode == MODE_R2R) *bytes = (double)p.d.batch * 2.0 * p.d.r2r_n * (double)pl->precision;
  }
  return GFFT_OK;
}

/* Fuse the 3/2-rule truncation (forward kinds) or zero-padding (backward kinds) of
 * libfft.py:263-311 into a single-axis plan: the truncated array (n_keep entries along the axis:
 * N for a complex axis, N/2+1 for the real half-axis) becomes the plan's output (input).
 * Returns GFFT_ERR_UNSUPPORTED, leaving the plan unchanged, when the plan is not one
 * register-kernel pass (the caller then uses gfft_truncate / gfft_pad). */
int gfft_plan_set_truncation(gfft_plan pl, int64_t n_keep) {
  if (!pl) return fail(GFFT_ERR_INVALID, "null plan");
  if (pl->passes.size() != 1 || pl->axes.size() != 1 || pl->fused3)
    return fail(GFFT_ERR_UNSUPPORTED, "truncation fuses into single-pass plans only");
  Pass &p = pl->passes[0];
  if (p.d.mode == MODE_R2C_H || p.d.mode == MODE_C2R_H) {
    // packed-real rows: the truncating store / zero-padding load act on the half spectrum
    // (entries 0 .. n_keep-1 of the p.d.n + 1; strides are in complex entries, inner == 1)
    const int64_t full_h = (int64_t)p.d.n + 1;
    if (n_keep < 1 || n_keep > full_h) return fail(GFFT_ERR_INVALID, "bad truncated length");
    if (n_keep == full_h) return GFFT_OK;        // nothing to cut: the plain plan is the answer
    if (!fused_pad_ok(p.d.n)) return fail(GFFT_ERR_UNSUPPORTED, "no fused truncation kernels for this length in this build");
    const bool fwd_h = p.d.mode == MODE_R2C_H;
    const double lines_h = (double)p.d.batch, esz_h = 2.0 * pl->precision;
    if (fwd_h) { p.d.tr_dir = 1; p.d.out_os = n_keep; } else { p.d.tr_dir = 2; p.d.in_os = n_keep; }
    pl->bytes += lines_h * ((double)n_keep - (double)full_h) * esz_h;
    p.d.tr_n = p.d.tr_N = (int)n_keep;
    p.d.tr_even = (n_keep % 2 == 0) ? 1 : 0;
    return GFFT_OK;
  }
  if (p.kind != PK_FFT || !p.regk || p.d.mid != 1 || p.d.tw_hi || !fused_pad_ok(p.d.n))
    return fail(GFFT_ERR_UNSUPPORTED, "truncation fuses into register-kernel passes only");
  const int axis = pl->axes[0];
  const bool fwd = pl->kind == GFFT_C2C_FORWARD || pl->kind == GFFT_R2C;
  const bool real = pl->kind == GFFT_R2C || pl->kind == GFFT_C2R;
  const int64_t full = real ? p.d.n / 2 + 1 : p.d.n;
  if (n_keep < 1 || n_keep > full) return fail(GFFT_ERR_INVALID, "bad truncated length");
  int64_t inner = 1;
  for (int i = axis + 1; i < pl->ndims; ++i) inner *= pl->sizes_in[i];
  const double esz = 2.0 * pl->precision;
  const double lines = (double)p.d.batch;
  if (fwd) {
    p.d.tr_dir = 1;
    p.d.out_os = n_keep * inner;
    pl->bytes -= lines * (double)full * esz;
  } else {
    p.d.tr_dir = 2;
    p.d.in_os = n_keep * inner;
    pl->bytes -= lines * (double)full * esz;
  }
  pl->bytes += lines * (double)n_keep * esz;
  p.d.tr_n = (int)n_keep;
  p.d.tr_N = (int)n_keep;
  p.d.tr_even = (n_keep % 2 == 0) ? 1 : 0;
  return GFFT_OK;
}

/* Packed-layout adapters (see include/gfft.h).  side 0: the plan reads its input from an
 * all-to-all receive buffer; side 1: it writes its output as an all-to-all send buffer.  Both are
 * the layout gfft_pack produces for `nblocks` equal blocks of the transformed axis:
 * [block][outer][n/nblocks][inner].  nblocks = 1 restores the natural layout. */
int gfft_plan_set_split(gfft_plan pl, int side, int nblocks) {
  if (!pl) return fail(GFFT_ERR_INVALID, "null plan");
  if (side != 0 && side != 1) return fail(GFFT_ERR_INVALID, "side must be 0 (input) or 1 (output)");
  if (pl->passes.size() != 1 || pl->axes.size() != 1 || pl->fused3)
    return fail(GFFT_ERR_UNSUPPORTED, "split layouts fuse into single-pass plans only");
  Pass &p = pl->passes[0];
  // (3 x 5 x 2^k lengths: their stages keep different numbers of values per thread, and a block boundary would have to
  // fall between the same thread slots on the load AND the store geometry -- natural layouts only, fft_mixv_*.hip)
  if (p.regk && mixv_supported(p.d.n)) return fail(GFFT_ERR_UNSUPPORTED, "split layouts: not for 3 x 5 x 2^k lengths");
  if ((p.d.mode == MODE_R2C_H && side == 1) || (p.d.mode == MODE_C2R_H && side == 0)) {
    // packed-real rows: the half-spectrum side as an all-to-all buffer of UNEVEN blocks (the
    // n/2 + 1 entries never divide evenly; pencil.py:5-9 deals the remainder to the first ranks)
    if (p.d.mid != 1 || p.d.inner != 1) return fail(GFFT_ERR_UNSUPPORTED, "split layout: packed-real rows only");
    // (with a fused truncation / zero padding the blocks are those of the KEPT entries)
    const int nh = p.d.tr_dir ? p.d.tr_n : p.d.n + 1;
    if (nblocks < 1 || nblocks > 8 || nblocks > nh) return fail(GFFT_ERR_UNSUPPORTED, "split layout: at most 8 blocks");
    if (nblocks == 1) {
      p.d.ub_p = 0;
      return GFFT_OK;
    }
    const int q = nh / nblocks, r = nh % nblocks;
    p.d.ub_p = nblocks;
    for (int b = 0; b <= nblocks; ++b) p.d.ub_start[b] = b * q + (b < r ? b : r);
    for (int b = nblocks + 1; b < 9; ++b) p.d.ub_start[b] = nh;
    p.d.ub_minw = q;
    p.d.ub_rows = p.d.batch;
    return GFFT_OK;
  }
  if (p.kind != PK_FFT || !p.regk || p.d.mid != 1 || p.d.tw_hi || p.d.mode != MODE_C2C)
    return fail(GFFT_ERR_UNSUPPORTED, "split layouts fuse into complex register-kernel passes only");
  const int64_t n = p.d.n, inner = p.d.inner, outer = p.d.batch / p.d.inner;
  int lg = 0;
  while ((1 << lg) < nblocks) ++lg;
  if (nblocks < 1 || (1 << lg) != nblocks) return fail(GFFT_ERR_UNSUPPORTED, "block count must be a power of two");
  if (p.d.tr_dir && side == (p.d.tr_dir == 1 ? 1 : 0)) {
    // the TRUNCATED side of a fused 3/2-rule truncation / zero padding: equal blocks of the kept
    // entries, addressed per entry (PassDesc::tr_jump) -- no tie to the kernel's thread layout
    const int64_t keep = p.d.tr_N;
    if (nblocks > 8 || keep % nblocks) return fail(GFFT_ERR_UNSUPPORTED, "block count does not divide the kept length");
    const int64_t per = keep / nblocks;
    if (nblocks > 1 && !is_pow2(per)) return fail(GFFT_ERR_UNSUPPORTED, "kept entries per block must be a power of two");
    int lgper = 0;
    while (((int64_t)1 << lgper) < per) ++lgper;
    if (nblocks > 1 && (double)(outer - 1) * per * inner * (nblocks - 1) >= 2147483648.0)
      return fail(GFFT_ERR_UNSUPPORTED, "blocked truncated side beyond 2^31 elements");      // (32-bit block offsets in the kernels)
    p.d.tr_lgper = nblocks > 1 ? lgper : 0;
    p.d.tr_jump = nblocks > 1 ? (outer - 1) * per * inner : 0;
    (side == 0 ? p.d.in_os : p.d.out_os) = per * inner;
    return GFFT_OK;
  }
  // whole thread slots per block: R = 4 (n = 16), 8 or 16 (other powers of two), 12 / 20 (3^b 2^k / 5^c 2^k)
  const int max_blocks = is_pow2(n) ? (n >= 32 ? 8 : 4) : 4;
  if (nblocks > max_blocks || n % nblocks) return fail(GFFT_ERR_UNSUPPORTED, "block count not supported for this length");
  const int64_t nb = n / nblocks;
  const int64_t os = nb * inner, jump = nblocks > 1 ? (outer - 1) * nb * inner : 0;
  // (what gfft_plan_create_guru would have recorded for the same layout: gfft_plan_set_tiles re-derives the block
  // jump from these)
  p.blocks[side] = nblocks;
  p.bstride[side] = nblocks > 1 ? outer * nb * inner : 0;
  if (side == 0) {
    p.d.in_os = os;
    p.d.in_jump = jump;
    p.d.in_lgp = lg;
  } else {
    p.d.out_os = os;
    p.d.out_jump = jump;
    p.d.out_lgp = lg;
  }
  return GFFT_OK;
}

int gfft_plan_create_guru(gfft_plan *plan, int precision, int kind, const gfft_iodim *dim, int howmany_rank,
                          const gfft_iodim *howmany, int in_blocks, int64_t in_block_stride, int out_blocks,
                          int64_t out_block_stride) {
  return gfft_plan_create_guru_padded(plan, precision, kind, dim, 0, howmany_rank, howmany, in_blocks, in_block_stride,
                                      out_blocks, out_block_stride);
}

int gfft_plan_create_guru_padded(gfft_plan *plan, int precision, int kind, const gfft_iodim *dim, int64_t n_keep,
                                 int howmany_rank, const gfft_iodim *howmany, int in_blocks, int64_t in_block_stride,
                                 int out_blocks, int64_t out_block_stride) {
  if (!plan || !dim || (howmany_rank > 0 && !howmany)) return fail(GFFT_ERR_INVALID, "null argument");
  *plan = nullptr;
  if (precision != GFFT_F32 && precision != GFFT_F64) return fail(GFFT_ERR_INVALID, "precision must be 4 or 8");
  if (kind != GFFT_C2C_FORWARD && kind != GFFT_C2C_BACKWARD) return fail(GFFT_ERR_UNSUPPORTED, "guru plans are complex-to-complex");
  if (howmany_rank < 0 || howmany_rank > 3) return fail(GFFT_ERR_UNSUPPORTED, "at most three batch dims");
  if (dim->n < 1) return fail(GFFT_ERR_INVALID, "bad transform length");
  int rc = check_device();
  if (rc) return rc;
  if (!regk_ok(dim->n, precision)) return fail(GFFT_ERR_UNSUPPORTED, "no single-pass register kernel for this length");
  // batch dims -> (outer, mid, inner): the last listed dim is the one adjacent columns run along
  gfft_iodim o{1, 0, 0}, m{1, 0, 0}, i{1, 0, 0};
  if (howmany_rank == 1) i = howmany[0];
  if (howmany_rank == 2) { o = howmany[0]; i = howmany[1]; }
  if (howmany_rank == 3) { o = howmany[0]; m = howmany[1]; i = howmany[2]; }
  if (o.n < 1 || m.n < 1 || i.n < 1) return fail(GFFT_ERR_INVALID, "bad batch length");
  const double batch = (double)o.n * (double)m.n * (double)i.n;
  if (batch >= 2147483648.0) return fail(GFFT_ERR_UNSUPPORTED, "batch exceeds 2^31");
  const int64_t n = dim->n;
  if (n_keep < 0 || n_keep > n) return fail(GFFT_ERR_INVALID, "bad kept length");
  const bool trunc = n_keep > 0 && n_keep < n;
  if (trunc && !fused_pad_ok(n)) return fail(GFFT_ERR_UNSUPPORTED, "no fused truncation kernels for this length in this build");
  const int tr_side = kind == GFFT_C2C_FORWARD ? 1 : 0;           // the side that holds the kept entries
  auto blocks_ok = [&](int nb, int side) {
    if (nb < 1 || (nb & (nb - 1))) return false;
    if (trunc && side == tr_side)                                // equal power-of-two blocks of the kept entries
      return nb <= 8 && n_keep % nb == 0 && (nb == 1 || is_pow2(n_keep / nb));
    const int max_blocks = is_pow2(n) ? (n >= 32 ? 8 : 4) : 4;    // whole thread slots per block (see gfft_plan_set_split)
    return nb <= max_blocks && n % nb == 0;
  };
  if (!blocks_ok(in_blocks, 0) || !blocks_ok(out_blocks, 1)) return fail(GFFT_ERR_UNSUPPORTED, "block count not supported for this length");
  gfft_plan_s *pl = new gfft_plan_s;
  pl->ndims = 0;
  pl->kind = kind;
  pl->precision = precision;
  pl->variant_rows = opts().variant_rows;
  pl->variant_cols = opts().variant_cols;
  pl->mixv_variant = opts().mixv_variant;
  pl->xcd_swizzle = opts().xcd_swizzle;
  Pass p;
  p.regk = true;
  p.cols = !(dim->is == 1 && dim->os == 1);
  p.logical_first = true;
  p.carries_scale = true;
  PassDesc &d = p.d;
  const bool inverse = kind == GFFT_C2C_BACKWARD;
  d.n = (int)n;
  d.mode = MODE_C2C;
  d.conj_in = d.conj_out = inverse ? 1 : 0;
  d.batch = o.n * m.n * i.n;
  d.mid = m.n;
  d.inner = i.n;
  d.in_os = o.is; d.in_ms = m.is; d.in_is = i.is; d.in_es = dim->is;
  d.out_os = o.os; d.out_ms = m.os; d.out_is = i.os; d.out_es = dim->os;
  d.scale = 1.0;
  auto lg2 = [](int v) { int l = 0; while ((1 << l) < v) ++l; return l; };
  if (in_blocks > 1 && !(trunc && tr_side == 0)) { d.in_lgp = lg2(in_blocks); d.in_jump = in_block_stride - (n / in_blocks) * dim->is; }
  if (out_blocks > 1 && !(trunc && tr_side == 1)) { d.out_lgp = lg2(out_blocks); d.out_jump = out_block_stride - (n / out_blocks) * dim->os; }
  if (trunc) {
    // fused 3/2-rule truncation (forward: store side) / zero padding (backward: load side), as
    // gfft_plan_set_truncation sets it on natural plans; blocks of the kept entries by PassDesc::tr_jump
    d.tr_dir = tr_side == 1 ? 1 : 2;
    d.tr_n = d.tr_N = (int)n_keep;
    d.tr_even = (n_keep % 2 == 0) ? 1 : 0;
    const int nb = tr_side == 1 ? out_blocks : in_blocks;
    if (nb > 1) {
      const int64_t per = n_keep / nb;
      d.tr_lgper = lg2((int)per);
      d.tr_jump = (tr_side == 1 ? out_block_stride - per * dim->os : in_block_stride - per * dim->is);
      if (std::fabs((double)d.tr_jump) * (nb - 1) >= 2147483648.0) {        // (32-bit block offsets in the kernels)
        delete pl;
        return fail(GFFT_ERR_UNSUPPORTED, "blocked truncated side beyond 2^31 elements");
      }
    }
  }
  p.blocks[0] = in_blocks;   p.bstride[0] = in_block_stride;
  p.blocks[1] = out_blocks;  p.bstride[1] = out_block_stride;
  rc = get_twiddles(n, precision, &d.tw);
  if (rc) { delete pl; return rc; }
  pl->passes.push_back(p);
  if (n > 1) pl->flops = 5.0 * (double)n * std::log2((double)n) * batch;
  pl->bytes = batch * (double)(n + (trunc ? n_keep : n)) * 2.0 * precision;
  *plan = pl;
  ++g_live_plans;
  return GFFT_OK;
}



/* The two local stages of a slab-decomposed transform as one plan (see include/gfft.h). */
int gfft_plan_create_guru2(gfft_plan *plan, int precision, int kind, const gfft_iodim *cols, const gfft_iodim *rows,
                           const gfft_iodim *planes, int cols_first, int in_blocks, int64_t in_block_stride,
                           int out_blocks, int64_t out_block_stride) {
  if (!plan || !cols || !rows || !planes) return fail(GFFT_ERR_INVALID, "null argument");
  *plan = nullptr;
  if (precision != GFFT_F32 && precision != GFFT_F64) return fail(GFFT_ERR_INVALID, "precision must be 4 or 8");
  if (kind != GFFT_C2C_FORWARD && kind != GFFT_C2C_BACKWARD) return fail(GFFT_ERR_UNSUPPORTED, "guru plans are complex-to-complex");
  const int64_t n1 = cols->n, n2 = rows->n, np = planes->n;
  if (n1 < 1 || n2 < 1 || np < 1) return fail(GFFT_ERR_INVALID, "bad length");
  if (rows->is != 1 || rows->os != 1) return fail(GFFT_ERR_UNSUPPORTED, "the row axis must be contiguous on both sides");
  if (in_blocks < 1 || out_blocks < 1) return fail(GFFT_ERR_INVALID, "bad block count");
  if (in_blocks > 1 && out_blocks > 1) return fail(GFFT_ERR_UNSUPPORTED, "blocks on one side only");
  int rc = check_device();
  if (rc) return rc;
  if (!regk_ok(n1, precision) || !regk_ok(n2, precision) || mixv_supported((int)n1))
    return fail(GFFT_ERR_UNSUPPORTED, "no single-pass register kernel for one of the lengths");
  for (int nb_ : {in_blocks, out_blocks}) {
    const int max_blocks = is_pow2(n1) ? (n1 >= 32 ? 8 : 4) : 4;           // whole thread slots per block (gfft_plan_set_split)
    if ((nb_ & (nb_ - 1)) || nb_ > max_blocks || n1 % nb_) return fail(GFFT_ERR_UNSUPPORTED, "block count not supported for this length");
  }
  if ((double)np * (double)n1 >= 2147483648.0 || (double)np * (double)n2 >= 2147483648.0) return fail(GFFT_ERR_UNSUPPORTED, "batch exceeds 2^31");
  const int64_t esz = 2 * (int64_t)precision;
  gfft_plan_s *pl = new gfft_plan_s;
  pl->ndims = 0;
  pl->kind = kind;
  pl->precision = precision;
  pl->variant_rows = opts().variant_rows;
  pl->variant_cols = opts().variant_cols;
  pl->mixv_variant = opts().mixv_variant;
  pl->xcd_swizzle = opts().xcd_swizzle;
  bool fused = false;
  rc = build_pair2d(pl, cols, n2, planes, kind == GFFT_C2C_BACKWARD, cols_first, in_blocks, in_block_stride, out_blocks, out_block_stride, false, &fused);
  if (rc) { delete pl; return rc; }
  pl->passes.back().carries_scale = true;
  const double elems = (double)np * (double)n1 * (double)n2;
  pl->flops = 5.0 * elems * std::log2((double)n1 * (double)n2);
  pl->bytes = 4.0 * elems * (double)esz;
  *plan = pl;
  ++g_live_plans;
  return GFFT_OK;
}

/* Tile-major layouts of exchange buffers (see include/gfft.h). */
int gfft_plan_set_tiles(gfft_plan pl, int side, int tile, int64_t tile_stride) {
  if (!pl) return fail(GFFT_ERR_INVALID, "null plan");
  if (side != 0 && side != 1) return fail(GFFT_ERR_INVALID, "side must be 0 (input) or 1 (output)");
  if (pl->passes.size() != 1 || pl->fused3) return fail(GFFT_ERR_UNSUPPORTED, "tile-major layouts: single-pass plans only");
  Pass &p = pl->passes[0];
  if (p.kind != PK_FFT || !p.regk || p.d.mode != MODE_C2C || p.d.tw_hi || p.d.tr_dir || mixv_supported(p.d.n))
    return fail(GFFT_ERR_UNSUPPORTED, "tile-major layouts: plain complex register-kernel passes only");
  int lg = 0;
  while ((1 << lg) < tile) ++lg;
  if (tile < 0 || (tile > 0 && ((1 << lg) != tile || tile < 2 || tile_stride < tile)))
    return fail(GFFT_ERR_INVALID, "tile must be a power of two >= 2 (0 switches the layout off)");
  PassDesc &d = p.d;
  if (p.cols) {
    // the adjacent columns (innermost batch dim) are tile-major
    if (side == 0) { d.in_ilg = tile ? lg : 0; d.in_iS = tile ? tile_stride : 0; }
    else { d.out_ilg = tile ? lg : 0; d.out_iS = tile ? tile_stride : 0; }
    return GFFT_OK;
  }
  // ROWS: the transformed axis itself is tile-major; thread slots advance by NT entries, which must be
  // whole tiles, and blocks must be whole tiles too
  const int64_t n = d.n;
  const int nb = p.blocks[side];
  if (tile) {
    const int nt = pow2_rows_nt(n, pl->precision);
    if (!nt || nt % tile || (n / nb) % tile)
      return fail(GFFT_ERR_UNSUPPORTED, "tile-major rows: the line's thread layout does not advance by whole tiles");
  }
  const int64_t es = 1;
  const int64_t per = n / nb;
  const int64_t span = tile ? (per / tile) * tile_stride : per * es;     // what one block advances the address by
  const int64_t jump = nb > 1 ? p.bstride[side] - span : 0;
  if (side == 0) { d.in_tlg = tile ? lg : 0; d.in_tS = tile ? tile_stride : 0; d.in_jump = jump; }
  else { d.out_tlg = tile ? lg : 0; d.out_tS = tile ? tile_stride : 0; d.out_jump = jump; }
  return GFFT_OK;
}

int gfft_plan_set_flat(gfft_plan pl, int64_t body_width, int64_t tail_offset, int64_t tail_row_stride) {
  if (!pl) return fail(GFFT_ERR_INVALID, "null plan");
  if (pl->passes.size() != 1 || pl->fused3) return fail(GFFT_ERR_UNSUPPORTED, "flat tiles: single-pass plans only");
  Pass &p = pl->passes[0];
  if (p.kind != PK_FFT || !p.regk || !p.cols || p.d.mode != MODE_C2C || p.d.tw_hi || p.d.tr_dir)
    return fail(GFFT_ERR_UNSUPPORTED, "flat tiles: plain complex strided register-kernel passes only");
  if (body_width < 0 || body_width > p.d.inner) return fail(GFFT_ERR_INVALID, "body width must lie in 0 .. columns per row");
  p.d.flat = 1;
  p.d.fl_bw = (body_width > 0 && body_width < p.d.inner) ? body_width : 0;
  p.d.fl_tail = tail_offset;
  p.d.fl_tail_ms = tail_row_stride;
  return GFFT_OK;
}

int gfft_plan_set_split_slabs(gfft_plan pl, int side, int nblocks, int64_t rows_per_slab, int tile) {
  if (!pl) return fail(GFFT_ERR_INVALID, "null plan");
  if (pl->passes.size() != 1 || pl->axes.size() != 1 || pl->fused3)
    return fail(GFFT_ERR_UNSUPPORTED, "split layouts fuse into single-pass plans only");
  Pass &p = pl->passes[0];
  if (!((p.d.mode == MODE_R2C_H && side == 1) || (p.d.mode == MODE_C2R_H && side == 0)))
    return fail(GFFT_ERR_UNSUPPORTED, "slab-wise split: the half-spectrum side of packed-real rows only");
  // Every failure below leaves the plan as it was; a repeated call starts from the natural row strides again (the
  // slab form multiplies them), and one block switches the slab form off altogether.
  const PassDesc saved = p.d;
  auto bail = [&](int code, const char *msg) { p.d = saved; return fail(code, msg); };
  if (p.d.ub_n1 > 0) {
    p.d.inner = 1;
    p.d.in_os = p.d.in_is;   p.d.in_is = 1;
    p.d.out_os = p.d.out_is; p.d.out_is = 1;
    p.d.ub_n1 = 0;
    p.d.ub_tlg = 0;
  }
  if (p.d.tr_dir || p.d.mid != 1 || p.d.inner != 1) return bail(GFFT_ERR_UNSUPPORTED, "slab-wise split: plain packed-real rows only");
  int lg = 0;
  while ((1 << lg) < tile) ++lg;
  if (tile < 2 || (1 << lg) != tile) return bail(GFFT_ERR_INVALID, "tile must be a power of two >= 2");
  if (rows_per_slab < 1 || p.d.batch % rows_per_slab) return bail(GFFT_ERR_INVALID, "rows per slab must divide the number of rows");
  // (the kernels address this layout with 32-bit element offsets)
  if ((double)p.d.batch * ((double)p.d.n + 1.0) >= 2147483648.0) return bail(GFFT_ERR_UNSUPPORTED, "slab-wise split: buffer beyond 2^31 entries");
  int rc = gfft_plan_set_split(pl, side, nblocks);
  if (rc) { p.d = saved; return rc; }
  if (nblocks == 1) return GFFT_OK;
  PassDesc &d = p.d;
  // rows of the batch as (slab, row in slab): the kernel's (o, i) indices
  d.inner = rows_per_slab;
  d.in_is = d.in_os;   d.in_os *= rows_per_slab;
  d.out_is = d.out_os; d.out_os *= rows_per_slab;
  d.ub_n1 = rows_per_slab;
  d.ub_tlg = lg;
  for (int b = 0; b < 8; ++b) d.ub_base[b] = b < nblocks ? d.ub_rows * (int64_t)d.ub_start[b] : 0;
  return GFFT_OK;
}

/* free the shared scratch buffers (they are otherwise kept for the life of the process) */
int gfft_scratch_release(void) { return scratch_pool().release(true); }

/* Did a fused launch of ANY plan give up a wait since the last look?  GFFT_OK, or GFFT_ERR_VOIDED once per event with the plan
 * named in gfft_last_error().  Does not synchronise: call it after the stream (or device) has been synchronised to learn
 * whether the results that synchronisation waited for are valid (include/gfft.h). */
int gfft_async_error(void) { return poll_async_error(nullptr); }
int gfft_plan_status(gfft_plan pl) {
  if (!pl) return fail(GFFT_ERR_INVALID, "null plan");
  return poll_async_error(pl);
}

int gfft_plan_destroy(gfft_plan pl) {
  if (!pl) return GFFT_OK;
  delete pl;
  if (--g_live_plans == 0) (void)scratch_pool().release(false);
  return GFFT_OK;
}

int gfft_plan_describe(gfft_plan pl, char *buf, size_t len) {
  if (!pl || !buf || !len) return fail(GFFT_ERR_INVALID, "null argument");
  std::string s;
  char line[256];
  const char *kn = pl->kind == GFFT_C2C_FORWARD ? "c2c-forward" : pl->kind == GFFT_C2C_BACKWARD ? "c2c-backward"
                   : pl->kind == GFFT_R2C ? "r2c" : "c2r";
  char sched[96] = "";
  if (pl->fused3 && pl->ws_tile) snprintf(sched, sizeof sched, " [3-D schedule: tile-major workspace, tiles of %d columns]", pl->ws_tile);
  else if (pl->fused3) snprintf(sched, sizeof sched, " [3-D schedule: padded-pitch workspace, rows %lld entries apart]", (long long)pl->ws_pitch);
  snprintf(line, sizeof line, "gfft plan: %s %s, %d dims, %zu passes%s\n", kn, pl->precision == 8 ? "f64" : "f32",
           pl->ndims, pl->passes.size(), sched);
  s += line;
  static const char *bufn[] = {"IN", "OUT", "WS", "FS", "AUX", "RING", "FS2"};
  for (const Pass &p : pl->passes) {
    if (p.kind == PK_FUSED2) {
      static const char *fk[] = {"rows -> strided", "strided -> rows", "four-step", "2-D planes: rows -> strided", "four-step: strided -> rows, transposed on store",
                                 "r2c rows -> strided", "strided -> c2r rows", "2-D planes: rows -> strided into blocks", "2-D planes: strided from blocks -> rows"};
      if (pl->fused_off)
        snprintf(line, sizeof line, "  pair (%s) n=%d then n=%d as two stand-alone passes (a fused launch gave up a wait)%s  %s -> %s\n",
                 fk[p.fused_kind], p.d.n, p.d2.n, p.carries_scale ? " [scale]" : "", bufn[p.src], bufn[p.dst]);
      else
      snprintf(line, sizeof line, "  fused pair (%s) n=%d then n=%d: %d planes, %d + %d tiles per plane, ring of %d slots x %lld KiB, one persistent launch%s  %s -> %s\n",
               fk[p.fused_kind], p.d.n, p.d2.n, p.fused.planes, p.fused.tiles_a, p.fused.tiles_b, p.fused.ring,
               (long long)(p.fused.slot_bytes >> 10), p.carries_scale ? " [scale]" : "", bufn[p.src], bufn[p.dst]);
      s += line;
      continue;
    }
    if (p.kind == PK_PLANE2D) {
      snprintf(line, sizeof line, "  planes on chip: rows n=%d then columns n=%d of %d planes held in LDS, one launch%s  %s -> %s\n", p.d.n, p.d2.n,
               p.fused.planes, p.carries_scale ? " [scale]" : "", bufn[p.src], bufn[p.dst]);
      s += line;
      continue;
    }
    if (p.kind != PK_FFT) {
      static const char *kinds[] = {"", "embed (chirp/zero-pad into AUX)", "multiply by B = FFT(chirp)", "extract (chirp, scale)"};
      snprintf(line, sizeof line, "  %s n=%lld Lw=%lld  %s -> %s\n", kinds[p.kind], (long long)p.pt.n,
               (long long)p.pt.Lw, bufn[p.src], bufn[p.dst]);
      s += line;
      continue;
    }
    snprintf(line, sizeof line, "  n=%d batch=%lld (mid=%lld inner=%lld) es_in=%lld es_out=%lld kernel=%s%s%s  %s -> %s\n",
             p.d.n, (long long)p.d.batch, (long long)p.d.mid, (long long)p.d.inner, (long long)p.d.in_es,
             (long long)p.d.out_es,
             (p.d.mode == MODE_R2C_H || p.d.mode == MODE_C2R_H) ? "regs-rows packed-real (n = complex length)"
             : p.regk ? (p.cols ? "regs-cols" : "regs-rows") : "generic",
             p.d.tw_hi ? " [four-step 1/2, fused twiddle]" : "", p.carries_scale ? " [scale]" : "",
             bufn[p.src], bufn[p.dst]);
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

int gfft_ps_curl(const void *d_u_hat, void *d_out, const void *d_k0, const void *d_k1, const void *d_k2,
                 int64_t n0, int64_t n1, int64_t n2, int precision, void *stream) {
  int rc = check_device();
  if (rc) return rc;
  if (!d_u_hat || !d_out || !d_k0 || !d_k1 || !d_k2 || n0 < 0 || n1 < 0 || n2 < 0 || (precision != 4 && precision != 8))
    return fail(GFFT_ERR_INVALID, "gfft_ps_curl: bad argument");
  HIP_TRY(launch_ps_curl(d_u_hat, d_out, d_k0, d_k1, d_k2, n0, n1, n2, precision, (hipStream_t)stream));
  return GFFT_OK;
}

int gfft_ps_cross(const void *d_a, const void *d_b, void *d_out, int64_t count, int precision, void *stream) {
  int rc = check_device();
  if (rc) return rc;
  if (!d_a || !d_b || !d_out || count < 0 || (precision != 4 && precision != 8))
    return fail(GFFT_ERR_INVALID, "gfft_ps_cross: bad argument");
  HIP_TRY(launch_ps_cross(d_a, d_b, d_out, count, precision, (hipStream_t)stream));
  return GFFT_OK;
}

int gfft_ps_project(void *d_du_hat, const void *d_u_hat, const void *d_k0, const void *d_k1, const void *d_k2,
                    int64_t n0, int64_t n1, int64_t n2, double nu, int precision, void *stream) {
  int rc = check_device();
  if (rc) return rc;
  if (!d_du_hat || !d_u_hat || !d_k0 || !d_k1 || !d_k2 || n0 < 0 || n1 < 0 || n2 < 0 || (precision != 4 && precision != 8))
    return fail(GFFT_ERR_INVALID, "gfft_ps_project: bad argument");
  HIP_TRY(launch_ps_project(d_du_hat, d_u_hat, d_k0, d_k1, d_k2, n0, n1, n2, nu, precision, (hipStream_t)stream));
  return GFFT_OK;
}

int gfft_ps_rk_stage(void *d_u, const void *d_u0, void *d_u1, const void *d_du, int64_t count, double cb,
                     double ca, int precision, void *stream) {
  int rc = check_device();
  if (rc) return rc;
  if ((d_u && !d_u0) || !d_u1 || !d_du || count < 0 || (precision != 4 && precision != 8))
    return fail(GFFT_ERR_INVALID, "gfft_ps_rk_stage: bad argument");
  HIP_TRY(launch_ps_rk(d_u, d_u0, d_u1, d_du, count, cb, ca, precision, (hipStream_t)stream));
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
  d.flat = opts().debug_flat;
  d.swizzle = opts().xcd_swizzle > 0 ? 1 : 0;
  if (!cols && (opts().debug_tile_side & 1)) { d.in_tlg = opts().debug_tile_lg; d.in_tS = opts().debug_tile_stride; }
  if (!cols && (opts().debug_tile_side & 2)) { d.out_tlg = opts().debug_tile_lg; d.out_tS = opts().debug_tile_stride; }
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
