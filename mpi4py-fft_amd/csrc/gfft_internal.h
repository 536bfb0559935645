// Internal declarations shared by the kernel translation units and the planner.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gfft {

// ordinal of the calling thread's current device, as an index into per-device state: function attributes, CU
// counts and the twiddle-table caches belong to ONE device, and a process may drive several
constexpr int kMaxDevices = 64;
inline int current_device() {
  int d = 0;
  if (hipGetDevice(&d) != hipSuccess) { (void)hipGetLastError(); d = 0; }
  return (d < 0 || d >= kMaxDevices) ? 0 : d;
}

// ---- complex helpers (device) ------------------------------------------------------------
template <typename T> struct cx { T x, y; };

template <typename T> __device__ __forceinline__ cx<T> operator+(cx<T> a, cx<T> b) { return {a.x + b.x, a.y + b.y}; }
template <typename T> __device__ __forceinline__ cx<T> operator-(cx<T> a, cx<T> b) { return {a.x - b.x, a.y - b.y}; }
template <typename T> __device__ __forceinline__ cx<T> cmul(cx<T> a, cx<T> b) {
  return {a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x};
}
// a * (-i)
template <typename T> __device__ __forceinline__ cx<T> mul_mi(cx<T> a) { return {a.y, -a.x}; }
template <typename T> __device__ __forceinline__ cx<T> cswap(cx<T> a) { return {a.y, a.x}; }

// ---- one 1-D pass over a row-major array -------------------------------------------------
// Column b of the batch decomposes as b = (o * mid + m) * inner + i; element e of that column
// sits at  base + o*os + m*ms + i*is + e*es  (strides in units of the buffer's element type:
// real scalars on the real side of r2c/c2r, complex otherwise).
// MODE_R2C_H / MODE_C2R_H: real transforms of even length 2 n along a CONTIGUOUS axis as one complex
// transform of length n on the packed line z[j] = x[2j] + i x[2j+1] plus a Hermitian pass in
// registers (fft_pow2_impl.h); the real side is addressed as complex pairs (strides in pairs).
enum PassMode { MODE_C2C = 0, MODE_R2C = 1, MODE_C2R = 2, MODE_R2R = 3, MODE_R2C_H = 4, MODE_C2R_H = 5 };

struct PassDesc {
  int n;          // logical transform length
  int mode;       // PassMode
  int conj_in;    // conjugate on load   (inverse transform = conj . forward . conj)
  int conj_out;   // conjugate on store
  int swizzle;    // XCD-contiguous tile order (speed only)
  // fused 3/2-rule adapters (register kernels): 0 none, 1 truncate on store, 2 zero-pad on load
  int tr_dir, tr_n, tr_N, tr_even;
  // ... with the TRUNCATED side being an all-to-all buffer of 2^k equal blocks of the kept entries
  // (complex axes; gfft_plan_set_split / gfft_plan_create_guru_padded): kept entry k of a line sits
  // tr_jump * (k >> tr_lgper) elements beyond its place in a contiguous line.  tr_jump = 0: contiguous.
  int tr_lgper;
  int64_t tr_jump;
  int64_t batch;  // number of columns = outer * mid * inner
  int64_t mid, inner;
  // strided (COLS) passes over rows with padding columns: of the `inner` adjacent columns only the
  // first inner_ld are read (the rest enter as zeros) and only the first inner_st are written
  // (0 = all).  Lets the passes inside a pitched workspace run on whole 128-byte lines -- a ragged
  // last tile of ONE column (513-wide half spectra) costs 16-byte partial-line writes that measured
  // 0.5 ms per 1024^3 pass, profiles/r02b_ragged_probe.txt -- while the caller's natural arrays are
  // touched only where they have data.
  int64_t inner_ld, inner_st;
  // strided (COLS) passes: tiles of adjacent columns run over the FLATTENED (mid, inner) index
  // J = m * inner + i instead of staying inside one row -- for a side whose rows lie back to back
  // (ms == inner * is) the tile's segments are then aligned whenever the array is, whatever the row
  // width (513-wide half spectra); the other side is addressed per lane through (m, i) = divmod(J, inner)
  int flat;
  int grid_cap;   // workgroups launched at most for this pass (0 = the library default, pow2_grid_cap())
  int out_pad;    // MODE_R2C_H: zero entries written after X[N] (fills the output row's last line)
  // Packed-real rows writing (r2c) / reading (c2r) an all-to-all buffer whose blocks are UNEVEN: the
  // N + 1 entries of the half spectrum dealt to ub_p ranks by the reference's block rule
  // (pencil.py:5-9), block b = entries [ub_start[b], ub_start[b+1]).  Entry e of row o (of ub_rows
  // rows in the buffer) lives at  ub_rows * start_b + o * w_b + (e - start_b):  the layout gfft_pack
  // produces for the cut axis being the last one.  0 = not in use.
  int ub_p;
  int ub_minw;            // width of the narrowest block (the kernels' group addressing needs it >= their threads per row)
  int ub_start[9];
  int64_t ub_rows;
  int64_t in_os, in_ms, in_is, in_es;
  int64_t out_os, out_ms, out_is, out_es;
  // packed-layout adapters (gfft_plan_set_split): the transform axis is cut into 2^lgp equal
  // blocks that lie `jump` elements further apart than consecutive entries would (the layout of
  // an all-to-all send / receive buffer); 0 / 0 = natural layout
  int in_lgp, out_lgp;
  int64_t in_jump, out_jump;
  // tile-major transform axis (ROWS mapping; exchange buffers laid out for the strided pass of the
  // NEXT stage, whose tile of 2^tlg adjacent columns then is one contiguous run): entry e of a line sits
  // (e >> tlg) * tS + (e & (2^tlg - 1)) elements from the line's base instead of e * es; 0 = off.
  // Thread slots e = t + q NT advance by NT, a multiple of the tile width, so the step stays uniform.
  int in_tlg, out_tlg;
  int64_t in_tS, out_tS;
  // tile-major COLUMNS (COLS mapping: the strided pass that reads / writes such a buffer): adjacent
  // column i of a row of the batch sits (i >> ilg) * iS + (i & (2^ilg - 1)) elements from the row's base
  // instead of i * is; 0 = off.  One shift-and-mask per lane and tile, nothing per element.
  int in_ilg, out_ilg;
  int64_t in_iS, out_iS;
  // flat tiles (see `flat`) over rows stored as a BODY of fl_bw columns plus the remaining columns as
  // a narrow array of their own (odd-width half spectra in exchange buffers: 513 = 512 + 1): on the
  // INPUT side column i >= fl_bw of row m is read from fl_tail + m * fl_tail_ms + (i - fl_bw).  0 = off.
  int64_t fl_bw, fl_tail, fl_tail_ms;
  // Uneven blocks stored slab by slab (ub_n1 > 0; gfft_plan_set_split_slabs): the rows of the batch are
  // (slab o, row i) with ub_n1 rows per slab; block b of a slab is its rows cut to the BODY columns
  // -- w_b rounded down to whole 2^ub_tlg entries, so that every row is whole 128-byte lines -- [row][body],
  // followed by the leftover columns [row][w_b - body]; the slabs of one block lie back to back from
  // element ub_base[b].
  int ub_tlg;
  int64_t ub_n1;
  int64_t ub_base[8];
  // MODE_R2R in the register kernels (DCT / DST kinds as one complex transform of the logical
  // length n = 2 r2r_n, see plan.cpp plan_r2r_line): the load reads REAL entry j = e - r2r_pos0
  // (zero outside [0, r2r_n)) times r2r_pre[j]; the store writes the REAL value
  // Re(r2r_post[k] * Z[e]) to entry k = e - r2r_idx0 when it lies in [0, r2r_n).  Strides are in
  // real elements on both sides.
  const void *r2r_pre, *r2r_post;
  int r2r_n, r2r_pos0, r2r_idx0;
  double scale;           // applied on store
  const void *tw;         // cx<real>[n]: exp(-2 pi i k / n)
  const void *rtw;        // MODE_R2C_H / MODE_C2R_H: cx<real>[n]: exp(-2 pi i k / (2 n)), k < n
  // optional four-step twiddle: output element k of a column with mid index m is multiplied
  // by W_big^(m*k), W_big = exp(-2 pi i / big_n), m*k < big_n <= 2^24;
  // factored as hi[(m*k) >> tw_L] * lo[(m*k) & (2^tw_L - 1)]
  const void *tw_hi, *tw_lo;
  int64_t big_n;
  int tw_L;
};

// pointwise helpers of the embedding fallbacks (Bluestein, long real transforms): lines of the
// array [outer][nin|nout][inner] are copied into / out of a complex scratch [outer][Lw][inner]
struct PointDesc {
  int64_t outer, inner;
  int64_t n;          // logical transform length
  int64_t nin, nout;  // entries per line on the input / output side (n, or n/2+1 on a half spectrum)
  int64_t Lw;         // line length in the scratch (n, or Bluestein's M)
  int mode;           // PassMode
  int conj;           // conjugate (embed: on load; extract: on store)
  const void *chirp;  // cx<real>[n]: exp(-i pi j^2 / n), or null
  const void *B;      // cx<real>[Lw]: FFT of the wrapped conjugate chirp (PK_MULB)
  // MODE_R2R (DCT / DST kinds as one complex transform of the logical length Lw, plan.cpp
  // plan_r2r_line): embed writes z[pos0 + j] = pre[j] * x[j], zeros elsewhere; extract writes
  // y[k] = Re(post[k] * Z[idx0 + k]).  pre / post: cx<real>[n]
  const void *pre, *post;
  int pos0, idx0;
};

struct Factors {
  int count;
  int r[24];
};

// launchers implemented in the kernel translation units; all return hipError_t
hipError_t launch_generic(const PassDesc &d, const Factors &f, int precision, const void *in,
                          void *out, hipStream_t s);
// max transform length the generic LDS kernel accepts for this precision
int generic_max_n(int precision);

// A strided (COLS) pass touches its 128-byte lines WHOLE on both sides: entries of a column, rows (m), outer slabs (o) and the blocks of an all-to-all
// buffer are whole lines apart (adjacent columns are adjacent elements; tile-major and flat layouts are aligned by construction).  What decides
// between non-temporal and plain streams: write-around stores of PARTIAL lines cost up to 65 % (profiles/r06_cols_nt_probe.txt) -- also where only
// the ROW pitch is odd (a 513-wide natural array behind a 512-column body: the far stage of config C5's odd rank column, 2.29 -> 2.44 ms).
inline bool strided_lines_whole(const PassDesc &d, int esz) {
  auto ok = [&](int64_t s) { return (s * esz) % 128 == 0; };
  if (!ok(d.in_es) || !ok(d.out_es)) return false;
  const int64_t rows = d.mid * d.inner > 0 ? d.batch / (d.mid * d.inner) : 1;
  if (d.mid > 1 && !d.flat && (!ok(d.in_ms) || !ok(d.out_ms))) return false;
  if (rows > 1 && (!ok(d.in_os) || !ok(d.out_os))) return false;
  if ((d.in_lgp && !ok(d.in_jump)) || (d.out_lgp && !ok(d.out_jump))) return false;
  return true;
}

// fast power-of-two kernels; returns false if (n, precision) has no instantiation
bool pow2_supported_f64(int n);
bool pow2_supported_f32(int n);
hipError_t launch_pow2_f64(const PassDesc &d, bool cols, int variant, const void *in, void *out, hipStream_t s);
hipError_t launch_pow2_f32(const PassDesc &d, bool cols, int variant, const void *in, void *out, hipStream_t s);
// MODE_R2R passes (power-of-two logical lengths 64 ... 4096)
bool pow2_r2r_supported(int n);
hipError_t launch_pow2_r2r_f64(const PassDesc &d, bool cols, const void *in, void *out, hipStream_t s);
hipError_t launch_pow2_r2r_f32(const PassDesc &d, bool cols, const void *in, void *out, hipStream_t s);
int pow2_grid_cap();
// two dependent fp64 passes fused into one persistent launch (fft_fused_f64.hip; fft_pow2_impl.h
// fft_fused2_kernel).  kind: FUSED_ROWS_COLS = [rows, n] -> [strided, n], FUSED_COLS_ROWS the reverse,
// FUSED_FOURSTEP = the two passes of a four-step transform of length n * n
struct FusedDesc {
  int planes, tiles_a, tiles_b, ring, lag;
  int defer;                         // 1: an A tile's counter is settled from inside the workgroup's next tile (fft_fused2_kernel)
  int group;                         // tiles per ticket (divides tiles_a and tiles_b): fewer tickets, counters and acknowledgement waits per byte
  int64_t a_in_plane, b_out_plane;   // BYTES from one plane to the next on A's input / B's output side
  int64_t slot_bytes;
  unsigned debug;                    // developer aid (GFFT_FUSE2_DEBUG): 2 tickets only, 3 + waits, 4 + A tiles, 5 + B tiles instead
  unsigned wait_ticks;               // 100 MHz wall-clock ticks a workgroup waits for a counter before it voids the launch (0: at once)
  unsigned plan_id;                  // what a launch that gave up writes to *host_flag
  unsigned *host_flag;               // pinned host word the library polls (plan.cpp poll_async_error); may be null
  unsigned *ctr;                     // [0] ticket, [1] launch void (a wait gave up), [16 + p] A tiles of plane p stored, [16 + planes + p] B tiles done
};
enum FusedKind { FUSED_ROWS_COLS = 0, FUSED_COLS_ROWS = 1, FUSED_FOURSTEP = 2, FUSED_PLANES_2D = 3, FUSED_FOURSTEP_ROWS = 4,   // (3: the kernels of 0; 4: four-step with a row second pass)
                 // real transforms (fft_fused_real_f64.hip): [packed-real r2c rows -> strided] on the contiguous planes of a 3-D
                 // r2c schedule, and [strided -> packed-real c2r rows] of the c2r schedule
                 FUSED_R2C_PLANES = 5, FUSED_COLS_C2R = 6,
                 // the two LOCAL stages of a slab-decomposed transform, plane by plane, with the all-to-all buffer addressed by
                 // the pair itself (gfft_plan_create_guru2): [rows -> strided, output in blocks] forward, [strided, input in
                 // blocks -> rows] backward -- the kernels of 3 / 1 with the block jump of the transformed axis kept (FLAGS 65536 / 32768)
                 FUSED_PLANES_2D_B = 7, FUSED_PLANES_CR_B = 8 };
// variant: 1 = the default kernels (32 values per thread, one exchange, one 512-thread workgroup per CU); 3 = the round-3
// kernels (16 values per thread, two exchanges, 1024 threads); 2 / 4 = (make VARIANTS=1) 8 lines per tile, two workgroups per CU
extern int g_fuse2_n512;           // option fuse2_n512: the n = 512 pairs (fft_fused_f64.hip)
extern int g_fuse2_mixed;          // option fuse2_mixed: pairs on planes of 512 x 1024 / 1024 x 512 points (fft_fused_f64.hip)
extern int g_fuse2_f32_n512;       // option fuse2_f32_n512: the complex64 n = 512 pairs (fft_fused_f32.hip)
extern int g_fuse2_mixv;           // option fuse2_mixv: the 3-D pair on n = 960 / 896 (fft_fused_f64.hip)
extern int g_c2r_2048;             // option c2r_2048: the c2r pair on rows of 2048 reals (fft_fused_real_f64.hip)
bool fused2_supported_f64(int kind, int variant, int n_a, int n_b);
int fused2_tiles_f64(int kind, int variant, const PassDesc &dA, const PassDesc &dB, int *tiles_a, int *tiles_b);
// dev_descs: {dA, dB} in device memory (uploaded when the plan was made; the scale factors travel as arguments)
hipError_t launch_fused2_f64(int kind, int variant, const PassDesc &dA, const PassDesc &dB, const PassDesc *dev_descs,
                             const FusedDesc &f, const void *in, void *ring, void *out, hipStream_t s);
// ... complex64 (fft_fused_f32.hip; on by default: option fuse2_f32 = 1, 0 switches them off; measured, see there)
bool fused2_supported_f32(int kind, int n_a, int n_b);
int fused2_tiles_f32(int kind, const PassDesc &dA, const PassDesc &dB, int *tiles_a, int *tiles_b);
hipError_t launch_fused2_f32(int kind, const PassDesc &dA, const PassDesc &dB, const PassDesc *dev_descs, const FusedDesc &f,
                             const void *in, void *ring, void *out, hipStream_t s);
// ... the two real kinds (n_a / n_b: the passes' lengths -- COMPLEX length of the packed-real rows)
bool fused2_real_supported_f64(int kind, int n_a, int n_b);
int fused2_real_tiles_f64(int kind, const PassDesc &dA, const PassDesc &dB, int *tiles_a, int *tiles_b);
hipError_t launch_fused2_real_f64(int kind, const PassDesc &dA, const PassDesc &dB, const PassDesc *dev_descs, const FusedDesc &f,
                                  const void *in, void *ring, void *out, hipStream_t s);
bool fused2_real_supported_f32(int kind, int n_a, int n_b);
int fused2_real_tiles_f32(int kind, const PassDesc &dA, const PassDesc &dB, int *tiles_a, int *tiles_b);
hipError_t launch_fused2_real_f32(int kind, const PassDesc &dA, const PassDesc &dB, const PassDesc *dev_descs, const FusedDesc &f,
                                  const void *in, void *ring, void *out, hipStream_t s);

// packed-real row kernels (fft_real_*.hip): d.n = complex length = half the real length
bool real_half_supported(int n_complex);
hipError_t launch_real_half_f64(const PassDesc &d, int variant, const void *in, void *out, hipStream_t s);
hipError_t launch_real_half_f32(const PassDesc &d, int variant, const void *in, void *out, hipStream_t s);
// ... for complex lengths 3^b 2^k / 5^c 2^k (fft_real_mix_*.hip, generated)
bool real_half_mix_supported(int n_complex);
hipError_t launch_real_half_mix_f64(const PassDesc &d, const void *in, void *out, hipStream_t s);
hipError_t launch_real_half_mix_f32(const PassDesc &d, const void *in, void *out, hipStream_t s);
// lengths 3^b * 2^k handled by the same register-resident kernel with R = 12 (fft_mix3_*.hip)
bool mix3_supported(int n);
hipError_t launch_mix3_f64(const PassDesc &d, bool cols, const void *in, void *out, hipStream_t s);
hipError_t launch_mix3_f32(const PassDesc &d, bool cols, const void *in, void *out, hipStream_t s);
// lengths 3 x 5 x 2^k and neighbours, stages of unequal width (fft_mixv_*.hip): plain complex passes only
bool mixv_supported(int n);
hipError_t launch_mixv_f64(const PassDesc &d, bool cols, int variant, const void *in, void *out, hipStream_t s);
hipError_t launch_mixv_f32(const PassDesc &d, bool cols, int variant, const void *in, void *out, hipStream_t s);
hipError_t launch_real_half_mixv_f64(const PassDesc &d, const void *in, void *out, hipStream_t s);     // packed-real rows, d.n = complex length
hipError_t launch_real_half_mixv_f32(const PassDesc &d, const void *in, void *out, hipStream_t s);
// lengths 5^c * 2^k with R = 20 (fft_mix5_*.hip)
bool mix5_supported(int n);
hipError_t launch_mix5_f64(const PassDesc &d, bool cols, const void *in, void *out, hipStream_t s);
hipError_t launch_mix5_f32(const PassDesc &d, bool cols, const void *in, void *out, hipStream_t s);

// both passes of small 2-D planes in one launch, the plane in LDS (fft_plane2d.hip): dA = rows of one plane -> LDS, dB = columns LDS -> out
bool plane2d_supported(int n, int precision);
int plane2d_pitch(int n);
hipError_t launch_plane2d(const PassDesc &dA, const PassDesc &dB, int precision, int planes, int64_t in_plane, int64_t out_plane,
                          const void *in, void *out, hipStream_t s);

hipError_t launch_pack(const void *src, void *dst, int64_t outer, int64_t naxis, int64_t inner,
                       int nparts, int itemsize, bool unpack, hipStream_t s);
hipError_t launch_trunc(const void *src, void *dst, int64_t outer, int64_t npad,
                        int64_t ntrunc, int64_t inner, int is_real, int precision, double scale,
                        bool pad_direction, hipStream_t s);
hipError_t launch_embed(const PointDesc &p, int precision, const void *in, void *scratch, hipStream_t s);
hipError_t launch_mulb(const PointDesc &p, int precision, void *scratch, hipStream_t s);
hipError_t launch_extract(const PointDesc &p, int precision, const void *scratch, void *out, double scale, hipStream_t s);
hipError_t launch_scale(void *data, int64_t count, int precision, double scale, hipStream_t s);
hipError_t launch_ps_curl(const void *u, void *out, const void *k0, const void *k1, const void *k2, int64_t n0,
                          int64_t n1, int64_t n2, int precision, hipStream_t s);
hipError_t launch_ps_cross(const void *a, const void *b, void *out, int64_t count, int precision, hipStream_t s);
hipError_t launch_ps_project(void *du, const void *u, const void *k0, const void *k1, const void *k2, int64_t n0,
                             int64_t n1, int64_t n2, double nu, int precision, hipStream_t s);
hipError_t launch_ps_rk(void *u, const void *u0, void *u1, const void *du, int64_t count, double cb, double ca,
                        int precision, hipStream_t s);
extern int g_copy_nt;
hipError_t launch_copy(const void *src, void *dst, size_t bytes, hipStream_t s);
hipError_t launch_tile_copy(const void *src, void *dst, int64_t outer, int64_t n, int64_t inner,
                            int tcols, hipStream_t s);

}  // namespace gfft
