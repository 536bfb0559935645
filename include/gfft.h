/* gfft.h -- C ABI of libgfft.so, the MI355X (gfx950) engine behind the PFFT hot path.
 *
 * Drop-in boundary: these entry points are what mpi4py-fft's native layer binds today
 * (Cython `fftw_xfftn.FFT` over `fftw_planxfftn`, plus what MPI's datatype engine does inside
 * `Alltoallw`), re-expressed for device memory:
 *
 *   gfft_plan_create    <- fftw_planxfftn()            mpi4py_fft/fftw/fftw_planxfftn.h:9-17, .c:10-77
 *   gfft_execute        <- fftw_execute_dft{,_r2c,_c2r}  mpi4py_fft/fftw/fftw_xfftn.pyx:29-48,291-292
 *   gfft_plan_destroy   <- fftw_destroy_plan()         mpi4py_fft/fftw/fftw_xfftn.pyx:162-163
 *   gfft_plan_describe  <- fftw_print_plan()           mpi4py_fft/fftw/fftw_xfftn.pyx:173-175
 *   gfft_pack/unpack    <- Create_subarray + Alltoallw's pack/unpack   mpi4py_fft/pencil.py:12-29,182,200
 *   gfft_truncate/pad   <- FFTBase._truncation_forward/_padding_backward  mpi4py_fft/libfft.py:263-311
 *   gfft_scale          <- `output_array *= M`         mpi4py_fft/libfft.py:412-413, fftw_xfftn.pyx:293-294
 *
 * Conventions: every function returns 0 (GFFT_OK) or a negative gfft_status; nothing throws
 * across the ABI.  All pointers named d_* are DEVICE pointers, borrowed (never owned or freed by
 * the library).  All work is enqueued asynchronously on `stream` (a hipStream_t passed as void*;
 * NULL = the default stream).  Sizes and strides are 64-bit (the reference's C planner uses
 * `int`, fftw_planxfftn.c:11-22; 1024^3 is its edge).  Arrays are C-contiguous (row-major), the
 * only layout the reference plans for (fftw_planxfftn.c:25-30).  Caller is single-threaded per
 * plan.  No host fallback exists: without a HIP device every compute entry point returns
 * GFFT_ERR_NO_DEVICE.
 *
 * Scratch and re-entrancy: plans that need a workspace (3-D all-axes plans, multi-axis c2r, four-step
 * and Bluestein lengths) carve it from ONE buffer per (calling thread, stream), shared by every
 * plan that thread executes on that stream and grown -- with a synchronisation of that stream --
 * when a larger request arrives (gfft_scratch_release frees them; so does destroying the last
 * plan of the process).  Consequences: (1) a thread may
 * run any sequence of plans on a stream; (2) two executions issued by one thread that may overlap
 * in time must be on different streams; (3) the first
 * gfft_execute of the largest plan on a stream allocates, so run each plan once on the stream
 * before capturing it into a HIP graph -- after that gfft_execute and the pointwise entries only
 * enqueue kernels (no allocation, no synchronisation); (4) a workspace handed out during a capture is
 * baked into the graph, so the library pins it: neither a larger plan arriving later nor the last plan
 * being destroyed frees it -- only gfft_scratch_release() does, and the caller must not replay such a
 * graph after calling it.
 */
#ifndef GFFT_H
#define GFFT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gfft_plan_s *gfft_plan;

typedef enum {
  GFFT_OK = 0,
  GFFT_ERR_INVALID = -1,     /* bad argument (shape/axes/kind mismatch)                */
  GFFT_ERR_UNSUPPORTED = -2, /* valid request the engine cannot plan (e.g. huge prime) */
  GFFT_ERR_NO_DEVICE = -3,   /* no HIP device / HIP runtime failure at init            */
  GFFT_ERR_HIP = -4,         /* a HIP call failed; see gfft_last_error()               */
  GFFT_ERR_NOMEM = -5,
  GFFT_ERR_VOIDED = -6       /* a launch that had already returned GFFT_OK failed on the device: gfft_async_error() */
} gfft_status;

/* transform kinds: same integers as the reference (fftw_planxfftn.c:3-8, utilities.pyx:7-26) */
enum { GFFT_C2C_FORWARD = -1, GFFT_C2C_BACKWARD = +1, GFFT_R2C = -2, GFFT_C2R = +2 };
/* real-to-real kinds: the integers of FFTW's fftw_r2r_kind as utilities.pyx:7-20 exposes them */
enum { GFFT_REDFT00 = 3, GFFT_REDFT01 = 4, GFFT_REDFT10 = 5, GFFT_REDFT11 = 6,
       GFFT_RODFT00 = 7, GFFT_RODFT01 = 8, GFFT_RODFT10 = 9, GFFT_RODFT11 = 10, GFFT_R2R = 100 };
/* precision of the real type: float (fftwf_* clone, setup.py:93-111) or double */
enum { GFFT_F32 = 4, GFFT_F64 = 8 };

const char *gfft_strerror(int status);
const char *gfft_last_error(void);          /* detail of the last failure on this thread */
int gfft_version(void);
int gfft_device_count(int *count);          /* hipGetDeviceCount; GFFT_ERR_NO_DEVICE if none */
int gfft_device_name(int device, char *buf, size_t len);
/* tunables consulted when a plan is created: "grid_cap", "variant_rows", "variant_cols",
 * "force_generic", "fused3", "profile" (also readable from the environment as GFFT_<UPPERCASE NAME>);
 * "fuse2" (1: pass pairs as one persistent launch through the Infinity Cache where a pair exists and pays,
 * 0: stand-alone passes), with "fuse2_ring" / "fuse2_lag" (slots of the hand-off ring /
 * planes the producer runs ahead; 0 = auto: about 96 MiB of lead and twice that of ring -- 12 / 6 planes of 16 MiB, 24 / 12 of
 * 8 MiB, 48 / 24 of 4 MiB --, launches with too few planes for that stay unfused), "fuse2_kinds" (bit mask of pair kinds:
 * 2 strided->rows, 4 / 16 four-step, 8 batched 2-D, 32 r2c rows->strided, 64 strided->c2r rows), "fuse2_f32" (1: complex64
 * pairs, 2: real fp32 pairs too), "fuse2_n512" (the n = 512 pairs: 0 off, 1 on 16 lines per tile, 2 [strided -> rows] on 32), "fuse2_mixed" (pairs on planes of 512 x 1024 / 1024 x 512 points), "fuse2_f32_n512" (the complex64 n = 512 pairs), "fuse2_wait_ms" (wall-clock limit of a wait inside a fused
 * launch before the launch voids itself, gfft_async_error below; default 2000); GFFT_FUSE2_DEBUG=1 prints a fused
 * launch's counters.  ("debug_tw_index" / "debug_tw_exp": TEST HOOK -- twiddle tables uploaded while debug_tw_exp > 0
 * carry one entry off by 10^-debug_tw_exp: what the rounding-level guards of tests/ must catch.) */
int gfft_set_option(const char *key, int value);

/* ---- serial multi-axis transform plan ------------------------------------------------
 * Same decomposition as fftw_planxfftn(): transform dims = `axes` (executed last-listed axis
 * first; for R2C/C2R the halved axis is axes[naxes-1], xfftn.py:231-232,309-315), every other
 * dim is a batch dim.  sizes_in/sizes_out are the full array shapes (they differ only along
 * axes[naxes-1] for R2C/C2R: n vs n/2+1).  R2C/C2R logical length n is taken from the real
 * side, as fftw_planxfftn.c:23 does.  */
int gfft_plan_create(gfft_plan *plan, int ndims, const int64_t *sizes_in, const int64_t *sizes_out,
                     int naxes, const int *axes, int kind, int precision);
/* Real-to-real plan: the `default:` branch of fftw_planxfftn (fftw_planxfftn.c:68-75,
 * fftw_plan_guru_r2r).  One kind per transformed axis (GFFT_REDFT00 .. GFFT_RODFT11 = DCT-I,
 * DCT-III, DCT-II, DCT-IV, DST-I, DST-III, DST-II, DST-IV in FFTW's unnormalised definitions);
 * input and output are real arrays of shape `sizes`; in-place execution is allowed. */
int gfft_plan_create_r2r(gfft_plan *plan, int ndims, const int64_t *sizes, int naxes, const int *axes,
                         const int *kinds, int precision);
/* out = scale * DFT(in).  d_in == d_out is allowed for C2C and real-to-real plans.  Every
 * out-of-place execution PRESERVES its input -- part of the contract: multi-axis C2R (where FFTW
 * destroys the input) routes its complex passes through the workspace, and the Python host relies
 * on it when it reads a caller's array in place (Transform.__call__). */
int gfft_execute(gfft_plan plan, const void *d_in, void *d_out, double scale, void *stream);
int gfft_plan_destroy(gfft_plan plan);
int gfft_scratch_release(void);           /* frees the shared per-stream workspaces, pinned ones included */
/* Errors of launches that have already returned GFFT_OK (execution is asynchronous): a fused pass-pair launch
 * whose workgroups waited longer than option "fuse2_wait_ms" (default 2000) for one another -- a device shared
 * with a long foreign kernel, a debugger -- voids itself instead of hanging or trapping.  The results of that plan's
 * last execution are invalid and the plan runs the pair as stand-alone launches from then on.  The event stays with
 * ITS plan until it is reported once, as GFFT_ERR_VOIDED with the plan named in gfft_last_error(), by whichever of
 * these looks first:
 *   gfft_plan_status(plan)  this plan's pending event;
 *   gfft_execute(plan, ..)  the same, before enqueueing anything (other plans' events do not refuse the call);
 *   gfft_async_error()      any plan's pending event (call until GFFT_OK to drain several).
 * None of them synchronises -- call after synchronising the stream to learn whether what was waited for is valid.
 * (The reference raises RuntimeError where FFTW fails to plan, mpi4py_fft/fftw/fftw_xfftn.pyx:152-153, and cannot
 * fail afterwards; this is the analogue for a failure mode only a persistent launch has.) */
int gfft_async_error(void);
int gfft_plan_status(gfft_plan plan);
/* Fuse FFTBase._truncation_forward / _padding_backward (libfft.py:263-311) into a single-axis plan:
 * afterwards gfft_execute writes (forward kinds) / reads (backward kinds) the TRUNCATED array,
 * n_keep entries along the axis (N on a complex axis, N/2+1 on the real half-axis), with the
 * reference's Nyquist rules.  GFFT_ERR_UNSUPPORTED = not fusable for this plan (use
 * gfft_truncate / gfft_pad); the plan is left unchanged. */
int gfft_plan_set_truncation(gfft_plan plan, int64_t n_keep);
/* The whole padded 3-D transform of a one-rank PFFT(padding=...) as ONE plan: what the reference runs
 * as three FFT objects, each followed (forward) by _truncation_forward or preceded (backward) by
 * _padding_backward (libfft.py:263-311, 408-422; chained by mpifft.py:68-73 with identity transfers).
 * `padded` = the transformed lengths = shape of the physical array (real for R2C / C2R); `kept` =
 * shape of the truncated spectral array (entries kept per axis: N on complex axes, N/2 + 1 on the
 * real half-axis, axis 2).  Forward kinds read the physical array and write the truncated one,
 * backward kinds the reverse; every pass carries its axis's truncating store / zero-padding load and
 * the intermediates live in the plan's pitched workspace (rows of the odd-width half spectrum start on
 * 128-byte lines there), never in arrays of the caller.  The input is preserved.
 * GFFT_ERR_UNSUPPORTED when a padded length has no single-pass kernel (caller keeps the per-axis
 * plans with gfft_plan_set_truncation). */
int gfft_plan_create_padded(gfft_plan *plan, const int64_t *padded, const int64_t *kept, int kind, int precision);
/* Fuse the pack / unpack side of Transfer (pencil.py:12-29,182,200: the subarray datatypes of
 * Alltoallw) into a single-axis complex plan.  side 1: gfft_execute writes its output directly
 * in the layout of the all-to-all SEND buffer for `nblocks` equal blocks of the transformed axis
 * -- what gfft_pack would produce from the natural output; side 0: it reads its input directly
 * from the RECEIVE buffer -- what gfft_unpack would consume.  nblocks = 1 restores the natural
 * layout.  GFFT_ERR_UNSUPPORTED (plan unchanged) when the plan is not one register-kernel pass or
 * nblocks is not a power of two <= 8 dividing the length.  Real transforms: the half-spectrum side
 * of an r2c (side 1) / c2r (side 0) plan along the contiguous last axis takes any nblocks <= 8 --
 * n/2 + 1 entries never split evenly, so the blocks follow the reference's block rule
 * (pencil.py:5-9) exactly as gfft_pack cuts them; other real plans return GFFT_ERR_UNSUPPORTED.
 * After gfft_plan_set_truncation the truncated side's blocks are blocks of the KEPT entries (complex
 * axes: equal blocks of a power of two of them; the real half-axis: the block rule over n_keep). */
int gfft_plan_set_split(gfft_plan plan, int side, int nblocks);
/* One batched 1-D complex transform with explicit strides: the form fftw_planxfftn() hands to
 * fftw_plan_guru_dft (fftw_planxfftn.c:25-57) -- `dim` is the transformed axis, `howmany` up to
 * three batch dims, slowest first; lengths and strides in elements, as fftw_iodim64 -- for callers
 * that run a stage of a distributed transform on sub-arrays and exchange buffers (the chunked,
 * stream-overlapped redistribution of mpi4py-fft_amd/pipeline.py).  Extension for exchange
 * buffers: the transformed axis may be stored as `in_blocks` / `out_blocks` equal blocks (a power
 * of two, 1 = contiguous) whose starts lie `*_block_stride` elements apart -- block b of the axis
 * is what rank b of the sub-communicator receives / sent.  kind: GFFT_C2C_FORWARD / _BACKWARD.
 * gfft_execute's d_in / d_out are the addresses of element 0.  GFFT_ERR_UNSUPPORTED when the
 * length has no single-pass register kernel or the block count does not fit its thread layout. */
typedef struct { int64_t n, is, os; } gfft_iodim;
int gfft_plan_create_guru(gfft_plan *plan, int precision, int kind, const gfft_iodim *dim, int howmany_rank,
                          const gfft_iodim *howmany, int in_blocks, int64_t in_block_stride, int out_blocks,
                          int64_t out_block_stride);
/* gfft_plan_create_guru with the 3/2-rule truncation (forward kind: on the store side) / zero padding
 * (backward kind: on the load side) of libfft.py:263-311 fused in, as gfft_plan_set_truncation fuses it
 * into natural plans: dim->n is the padded (transformed) length, `n_keep` the entries kept along the axis
 * on the truncated side -- the output of a forward plan, the input of a backward one -- whose stride
 * (dim->os / dim->is) and blocks then describe the TRUNCATED line: its blocks are equal blocks of the
 * kept entries (a power of two of them each), what the stage of a padded distributed transform
 * (mpifft.py:247-257 inflates the shape, :68-73 runs the chain) sends to / receives from its
 * sub-communicator.  n_keep = 0 or n: no truncation (= gfft_plan_create_guru). */
int gfft_plan_create_guru_padded(gfft_plan *plan, int precision, int kind, const gfft_iodim *dim, int64_t n_keep,
                                 int howmany_rank, const gfft_iodim *howmany, int in_blocks, int64_t in_block_stride,
                                 int out_blocks, int64_t out_block_stride);
/* The two LOCAL stages of a slab-decomposed transform as one plan.  The reference joins consecutive stages whose
 * redistribution stays on one rank with a whole-array self-Alltoallw (mpifft.py:324-331, pencil.py:168-183); the
 * staged form here elides that copy and runs two plans; this entry runs both transforms as ONE launch per
 * direction, plane by plane, the plane handed from the first pass to the second inside the Infinity Cache
 * (fft_pow2_impl.h fft_fused2_kernel) -- and the all-to-all buffer of the NEXT (previous) redistribution is
 * addressed by the strided pass itself, as gfft_plan_create_guru's blocks are.
 *   rows    the contiguous transformed axis (is = os = 1)
 *   cols    the strided transformed axis: n, is / os = distance between consecutive rows of a plane
 *   planes  the batch: n planes, is / os = distance between consecutive planes
 *   cols_first = 1: [cols, then rows] -- strided reads, whole rows written: the order the host uses in both directions
 *     (level with the other one in complex128, 9 % ahead in complex64: tools/stage_probe.py slab); 0: [rows, then cols].
 *   in_blocks / in_block_stride, out_blocks / out_block_stride: the cols axis stored as that many equal blocks whose
 *     starts lie that many elements apart (1 = contiguous) -- the receive buffer of the all-to-all that gathered that
 *     axis on the input side (backward direction), the send buffer of the one that scatters it on the output side
 *     (forward).  The strided pass adds the block jump to its step; the row pass places every row by its block.
 *     At most one side takes blocks.
 * kind: GFFT_C2C_FORWARD / _BACKWARD; gfft_execute's d_in / d_out are the addresses of element 0 and must not
 * overlap unless both sides have the same natural layout; the input is preserved otherwise.  Where the pair has
 * no fused kernels (planes other than 512 / 1024 points a side in fp64 -- equal or not --, 512 x 512 / 1024 x 1024 in fp32;
 * rows-first order outside the square fp64 / 1024 x 1024 fp32 pairs; too few planes for a hand-off ring) the
 * plan runs two stand-alone passes -- the first one carries the data across, the second works in place on the
 * output -- with the same results up to nothing: the arithmetic of a pass does not depend on its launch form.
 * GFFT_ERR_UNSUPPORTED when a length has no single-pass register kernel or the block count does not fit. */
int gfft_plan_create_guru2(gfft_plan *plan, int precision, int kind, const gfft_iodim *cols, const gfft_iodim *rows,
                           const gfft_iodim *planes, int cols_first, int in_blocks, int64_t in_block_stride,
                           int out_blocks, int64_t out_block_stride);
/* Layouts of INTERNAL exchange buffers (between two stages of a distributed transform; never of a
 * caller's array, which keeps the reference's C order, pencil.py:347-354).  All three act on one-pass
 * plans (gfft_plan_create_guru, or gfft_plan_create on one axis) and return GFFT_ERR_UNSUPPORTED,
 * leaving the plan as it was, when the plan's kernel cannot address the layout.
 *
 * gfft_plan_set_tiles: `side` (0 input, 1 output) is tile-major in tiles of `tile` elements (a power
 *   of two; 0 = off) that start `tile_stride` elements apart.  For a plan along a contiguous axis the
 *   TRANSFORMED axis is tiled: entry e of a line at (e / tile) * tile_stride + e % tile from the line's
 *   base (block starts -- gfft_plan_create_guru's in_blocks / out_blocks -- stay where they were).  For a
 *   strided plan the adjacent COLUMNS (last batch dim) are tiled: column i at (i / tile) * tile_stride +
 *   i % tile.  One describes the writer, the other the reader of the same buffer: the strided stage's
 *   tile of adjacent columns is then one contiguous run per workgroup (tools/stage_layout_probe.py).
 * gfft_plan_set_flat: a strided plan walks tiles over the FLATTENED (second-last, last) batch dims, so
 *   its accesses are line aligned on a side whose rows lie back to back whatever their width (513-wide
 *   half spectra).  body_width > 0: on the INPUT side a row is stored as `body_width` columns at the
 *   planned strides plus its remaining columns at tail_offset + row * tail_row_stride (elements from
 *   the outermost batch index's base) -- the layout a neighbouring stage writes with two plans.
 * gfft_plan_set_split_slabs: gfft_plan_set_split for the half-spectrum side of packed-real rows, the
 *   rows taken as slabs of `rows_per_slab`: block b of slab s is stored as its rows cut to the body
 *   columns (w_b rounded down to whole `tile` entries; `tile` = 256 bytes of complex values), [row][body],
 *   followed by the leftover columns [row][w_b mod tile]; a block's slabs lie back to back, the blocks
 *   one after the other (same message sizes as gfft_plan_set_split).  Rows of the 513-wide block of a
 *   2048-point real axis are then 512 entries = whole 128-byte lines, for this plan's stores and for
 *   the strided plan that reads them.  (Tile-major bodies -- what complex row plans write for free --
 *   cost the packed-real kernels 18-25 % in per-entry address arithmetic, profiles/r03_stage_probe*.txt.) */
int gfft_plan_set_tiles(gfft_plan plan, int side, int tile, int64_t tile_stride);
int gfft_plan_set_flat(gfft_plan plan, int64_t body_width, int64_t tail_offset, int64_t tail_row_stride);
int gfft_plan_set_split_slabs(gfft_plan plan, int side, int nblocks, int64_t rows_per_slab, int tile);
int gfft_plan_describe(gfft_plan plan, char *buf, size_t len);
/* flops (5 n log2 n per line, half for real) and algorithmic bytes (one read + one write of
 * the array per 1-D pass) of one execute, and the number of kernel launches it issues */
int gfft_plan_cost(gfft_plan plan, double *flops, double *bytes, int *launches);

/* ---- global-redistribution helpers (what Alltoallw's datatype engine does) ------------
 * An array of shape `shape[ndims]` is cut along `axis` into `nparts` blocks by the reference's
 * block rule (pencil.py:5-9).  pack: block i is copied, in row-major order of the sub-block, to
 * d_packed + offset_i (offset_i = itemsize * prod(other dims) * start_i), i.e. the send buffer of
 * an all-to-all.  unpack is the inverse (receive buffer -> array).  itemsize in bytes (4,8,16). */
int gfft_pack(const void *d_array, void *d_packed, int ndims, const int64_t *shape, int axis,
              int nparts, int itemsize, void *stream);
int gfft_unpack(const void *d_packed, void *d_array, int ndims, const int64_t *shape, int axis,
                int nparts, int itemsize, void *stream);

/* ---- 3/2-rule truncation / padding along one axis (libfft.py:263-311) ------------------
 * shapes differ only along `axis` (n_padded vs n_trunc entries).  is_real: the axis is the
 * Hermitian half-axis of an r2c transform.  Element type complex<precision>.  `scale` is applied
 * to every written element (fuses `*= M`).  */
int gfft_truncate(const void *d_padded, void *d_trunc, int ndims, const int64_t *shape_padded,
                  int axis, int64_t n_trunc, int is_real, int precision, double scale, void *stream);
int gfft_pad(const void *d_trunc, void *d_padded, int ndims, const int64_t *shape_padded,
             int axis, int64_t n_trunc, int is_real, int precision, void *stream);

/* d_data[i] *= scale for `count` real scalars of the given precision */
int gfft_scale(void *d_data, int64_t count, int precision, double scale, void *stream);

/* ---- the pseudo-spectral caller either side of the path ----
 * One-pass device versions of the pointwise steps examples/spectral_dns_solver.py:65-91 writes as
 * numpy expressions.  Vector fields are [3][n0][n1][n2] (the layout of newDistArray(fft, rank=1));
 * d_k0/1/2 are the local wavenumbers along each axis (n0, n1, n2 real scalars: the sparse form of
 * get_local_wavenumbermesh, :52-63), not array-sized meshes.
 *   gfft_ps_curl     out = 1j * (K x u_hat)                                   compute_curl, :76-80
 *   gfft_ps_cross    out = a x b, real fields, count = n0*n1*n2 per component   cross, :69-74
 *   gfft_ps_project  P = sum(du*K/|K|^2); du -= P*K; du -= nu*|K|^2*u_hat      compute_rhs, :88-90
 *   gfft_ps_rk_stage u = u0 + cb*du (skipped when d_u is NULL); u1 += ca*du; `count` real scalars   :112-116 */
int gfft_ps_curl(const void *d_u_hat, void *d_out, const void *d_k0, const void *d_k1, const void *d_k2,
                 int64_t n0, int64_t n1, int64_t n2, int precision, void *stream);
int gfft_ps_cross(const void *d_a, const void *d_b, void *d_out, int64_t count, int precision, void *stream);
int gfft_ps_project(void *d_du_hat, const void *d_u_hat, const void *d_k0, const void *d_k1, const void *d_k2,
                    int64_t n0, int64_t n1, int64_t n2, double nu, int precision, void *stream);
int gfft_ps_rk_stage(void *d_u, const void *d_u0, void *d_u1, const void *d_du, int64_t count, double cb,
                     double ca, int precision, void *stream);

/* ---- the wire of a global redistribution: RCCL over xGMI -------------------------------------
 * Replaces, for device buffers, what the reference gets from MPI on its Cartesian sub-communicators:
 *   gfft_comm_create / gfft_comm_split  <- MPI_Cart_create + MPI_Cart_sub   mpi4py_fft/pencil.py:64-93
 *   gfft_alltoallv / gfft_sendrecv      <- comm.Alltoallw(...)             mpi4py_fft/pencil.py:182-183,200-201
 * One process per GPU; a communicator is bound to the device current at its creation.  RCCL is
 * loaded at run time (an already loaded librccl is shared; GFFT_RCCL_LIB or gfft_rccl_load name a
 * specific one), so these entries return GFFT_ERR_UNSUPPORTED on a host without RCCL and nothing
 * else in the library depends on it.  The 128-byte unique id is produced on one rank and carried to
 * the others by whatever the host already has (MPI_Bcast, a torch.distributed store, a file).
 * All calls are collective where RCCL's are (create: all ranks; split: all ranks of the parent) and
 * every exchange is enqueued on `stream`: the host owns the stream and orders it against its compute
 * streams with events (gfft_event_record / gfft_stream_wait_event).  Error detail:
 * gfft_exchange_last_error(). */
typedef struct gfft_comm_s *gfft_comm;
typedef struct {
  void *ptr;        /* device address of the message */
  int64_t bytes;
  int peer;         /* rank in the communicator */
} gfft_msg;
#define GFFT_UNIQUE_ID_BYTES 128
int gfft_rccl_load(const char *path);                   /* NULL: default search */
int gfft_rccl_info(char *buf, size_t len);
const char *gfft_exchange_last_error(void);
int gfft_comm_get_unique_id(void *id128);
int gfft_comm_create(gfft_comm *comm, const void *id128, int nranks, int rank);
/* ranks passing the same color >= 0 form a group, ordered by key then parent rank; color < 0 joins
 * no group (*sub = NULL).  Sub-communicator of grid axis i: color = the other coordinates, key =
 * coordinate i (pencil.py:80-88). */
int gfft_comm_split(gfft_comm parent, int color, int key, gfft_comm *sub);
int gfft_comm_rank(gfft_comm comm, int *rank, int *size);
int gfft_comm_destroy(gfft_comm comm);
/* One grouped batch of point-to-point messages (ncclGroupStart .. ncclGroupEnd).  Messages
 * between two ranks match in list order; messages to oneself are device copies. */
int gfft_sendrecv(gfft_comm comm, int nsend, const gfft_msg *sends, int nrecv, const gfft_msg *recvs, void *stream);
/* MPI_Alltoallv on device buffers: send_counts[i] items at item offset send_displs[i] of d_send go
 * to rank i; recv_counts[j] items from rank j land at item offset recv_displs[j] of d_recv. */
int gfft_alltoallv(gfft_comm comm, const void *d_send, const int64_t *send_counts, const int64_t *send_displs,
                   void *d_recv, const int64_t *recv_counts, const int64_t *recv_displs, int itemsize, void *stream);
int gfft_stream_create(void **stream);                  /* non-blocking HIP stream */
int gfft_stream_destroy(void *stream);
int gfft_stream_wait_event(void *stream, void *event);
int gfft_event_create_untimed(void **event);            /* for ordering only (hipEventDisableTiming) */

/* ---- raw device helpers for non-torch hosts (the Python host uses torch for these) ---- */
int gfft_malloc(void **d_ptr, size_t bytes);
int gfft_free(void *d_ptr);
int gfft_memcpy_h2d(void *d_dst, const void *h_src, size_t bytes, void *stream);
int gfft_memcpy_d2h(void *h_dst, const void *d_src, size_t bytes, void *stream);
int gfft_memcpy_d2d(void *d_dst, const void *d_src, size_t bytes, void *stream);
int gfft_stream_synchronize(void *stream);

/* ---- measurement helpers used by bench.py (HIP events on the launch stream) ------------ */
int gfft_event_create(void **event);
int gfft_event_record(void *event, void *stream);
int gfft_event_elapsed_ms(void *start, void *stop, float *ms);   /* synchronises on `stop` */
int gfft_event_destroy(void *event);
/* per-pass kernel timing for the roofline report: set option "profile" to 1, execute, then read
 * the accumulated milliseconds of each pass (HIP events recorded on the execute stream) */
int gfft_plan_profile(gfft_plan plan, float *ms, int max_passes, int *executes);
int gfft_plan_pass_info(gfft_plan plan, int pass, char *buf, size_t len, double *algorithmic_bytes);
/* streaming-copy probe: dst[i] = src[i] over `bytes` (multiple of 16); the HBM ceiling quoted
 * beside roofline numbers */
int gfft_probe_copy(const void *d_src, void *d_dst, size_t bytes, void *stream);
/* strided-tile copy probe: the access pattern of a column pass without the arithmetic:
 * array [outer][n][inner] of 16-byte elements, tiles of `tcols` consecutive inner elements */
int gfft_probe_tile_copy(const void *d_src, void *d_dst, int64_t outer, int64_t n, int64_t inner,
                         int tcols, void *stream);
/* single-pass probe: run ONE power-of-two pass with explicit batch geometry and strides (element
 * units), bypassing the planner.  geom = {n, outer, mid, inner, in_os, in_ms, in_is, in_es, out_os,
 * out_ms, out_is, out_es}.  A measuring aid (tools/layout_probe.py), not part of the drop-in path. */
int gfft_debug_pass(const int64_t *geom, int precision, int cols, int variant, int inverse,
                    const void *d_in, void *d_out, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* GFFT_H */
