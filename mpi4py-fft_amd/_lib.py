"""ctypes binding of libgfft.so (include/gfft.h) and the engine object the host classes call.

There is exactly one product engine: :class:`HipEngine`, a thin veneer over the C ABI.  It has no
host fallback -- if the shared library is missing or no HIP device is visible every compute call
raises.  ``set_engine`` exists so that CPU-only unit tests of the *host logic* (pencil arithmetic,
exchange plans, gloo all-to-all) can inject a checker engine from tests/; nothing in this package
ever selects another engine on its own.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIBPATH = os.path.join(_HERE, 'libgfft.so')

C2C_FORWARD, C2C_BACKWARD, R2C, C2R = -1, 1, -2, 2

_lib = None


class GfftError(RuntimeError):
    pass


class IoDim(ctypes.Structure):          # gfft_iodim
    _fields_ = [('n', ctypes.c_int64), ('is_', ctypes.c_int64), ('os', ctypes.c_int64)]


class Msg(ctypes.Structure):            # gfft_msg
    _fields_ = [('ptr', ctypes.c_void_p), ('bytes', ctypes.c_int64), ('peer', ctypes.c_int)]


def _declare(lib):
    c = ctypes
    i64p = c.POINTER(c.c_int64)
    ip = c.POINTER(c.c_int)
    vp = c.c_void_p
    sigs = {
        'gfft_strerror': (c.c_char_p, [c.c_int]),
        'gfft_last_error': (c.c_char_p, []),
        'gfft_version': (c.c_int, []),
        'gfft_device_count': (c.c_int, [ip]),
        'gfft_device_name': (c.c_int, [c.c_int, c.c_char_p, c.c_size_t]),
        'gfft_set_option': (c.c_int, [c.c_char_p, c.c_int]),
        'gfft_plan_create': (c.c_int, [c.POINTER(vp), c.c_int, i64p, i64p, c.c_int, ip, c.c_int, c.c_int]),
        'gfft_plan_create_r2r': (c.c_int, [c.POINTER(vp), c.c_int, i64p, c.c_int, ip, ip, c.c_int]),
        'gfft_execute': (c.c_int, [vp, vp, vp, c.c_double, vp]),
        'gfft_plan_destroy': (c.c_int, [vp]),
        'gfft_scratch_release': (c.c_int, []),
        'gfft_plan_set_truncation': (c.c_int, [vp, c.c_int64]),
        'gfft_plan_create_padded': (c.c_int, [c.POINTER(vp), i64p, i64p, c.c_int, c.c_int]),
        'gfft_plan_set_split': (c.c_int, [vp, c.c_int, c.c_int]),
        'gfft_plan_create_guru': (c.c_int, [c.POINTER(vp), c.c_int, c.c_int, c.POINTER(IoDim), c.c_int, c.POINTER(IoDim),
                                            c.c_int, c.c_int64, c.c_int, c.c_int64]),
        'gfft_plan_create_guru_padded': (c.c_int, [c.POINTER(vp), c.c_int, c.c_int, c.POINTER(IoDim), c.c_int64, c.c_int,
                                                   c.POINTER(IoDim), c.c_int, c.c_int64, c.c_int, c.c_int64]),
        'gfft_plan_create_guru2': (c.c_int, [c.POINTER(vp), c.c_int, c.c_int, c.POINTER(IoDim), c.POINTER(IoDim), c.POINTER(IoDim),
                                             c.c_int, c.c_int, c.c_int64, c.c_int, c.c_int64]),
        'gfft_plan_set_tiles': (c.c_int, [vp, c.c_int, c.c_int, c.c_int64]),
        'gfft_plan_set_flat': (c.c_int, [vp, c.c_int64, c.c_int64, c.c_int64]),
        'gfft_plan_set_split_slabs': (c.c_int, [vp, c.c_int, c.c_int, c.c_int64, c.c_int]),
        'gfft_plan_describe': (c.c_int, [vp, c.c_char_p, c.c_size_t]),
        'gfft_plan_cost': (c.c_int, [vp, c.POINTER(c.c_double), c.POINTER(c.c_double), ip]),
        'gfft_pack': (c.c_int, [vp, vp, c.c_int, i64p, c.c_int, c.c_int, c.c_int, vp]),
        'gfft_unpack': (c.c_int, [vp, vp, c.c_int, i64p, c.c_int, c.c_int, c.c_int, vp]),
        'gfft_truncate': (c.c_int, [vp, vp, c.c_int, i64p, c.c_int, c.c_int64, c.c_int, c.c_int, c.c_double, vp]),
        'gfft_pad': (c.c_int, [vp, vp, c.c_int, i64p, c.c_int, c.c_int64, c.c_int, c.c_int, vp]),
        'gfft_scale': (c.c_int, [vp, c.c_int64, c.c_int, c.c_double, vp]),
        'gfft_ps_curl': (c.c_int, [vp, vp, vp, vp, vp, c.c_int64, c.c_int64, c.c_int64, c.c_int, vp]),
        'gfft_ps_cross': (c.c_int, [vp, vp, vp, c.c_int64, c.c_int, vp]),
        'gfft_ps_project': (c.c_int, [vp, vp, vp, vp, vp, c.c_int64, c.c_int64, c.c_int64, c.c_double, c.c_int, vp]),
        'gfft_ps_rk_stage': (c.c_int, [vp, vp, vp, vp, c.c_int64, c.c_double, c.c_double, c.c_int, vp]),
        'gfft_debug_pass': (c.c_int, [i64p, c.c_int, c.c_int, c.c_int, c.c_int, vp, vp, vp]),
        'gfft_malloc': (c.c_int, [c.POINTER(vp), c.c_size_t]),
        'gfft_free': (c.c_int, [vp]),
        'gfft_memcpy_h2d': (c.c_int, [vp, vp, c.c_size_t, vp]),
        'gfft_memcpy_d2h': (c.c_int, [vp, vp, c.c_size_t, vp]),
        'gfft_memcpy_d2d': (c.c_int, [vp, vp, c.c_size_t, vp]),
        'gfft_stream_synchronize': (c.c_int, [vp]),
        'gfft_event_create': (c.c_int, [c.POINTER(vp)]),
        'gfft_event_record': (c.c_int, [vp, vp]),
        'gfft_event_elapsed_ms': (c.c_int, [vp, vp, c.POINTER(c.c_float)]),
        'gfft_event_destroy': (c.c_int, [vp]),
        'gfft_plan_profile': (c.c_int, [vp, c.POINTER(c.c_float), c.c_int, ip]),
        'gfft_plan_pass_info': (c.c_int, [vp, c.c_int, c.c_char_p, c.c_size_t, c.POINTER(c.c_double)]),
        'gfft_probe_copy': (c.c_int, [vp, vp, c.c_size_t, vp]),
        'gfft_async_error': (c.c_int, []),
        'gfft_plan_status': (c.c_int, [vp]),
        'gfft_rccl_load': (c.c_int, [c.c_char_p]),
        'gfft_rccl_info': (c.c_int, [c.c_char_p, c.c_size_t]),
        'gfft_exchange_last_error': (c.c_char_p, []),
        'gfft_comm_get_unique_id': (c.c_int, [vp]),
        'gfft_comm_create': (c.c_int, [c.POINTER(vp), vp, c.c_int, c.c_int]),
        'gfft_comm_split': (c.c_int, [vp, c.c_int, c.c_int, c.POINTER(vp)]),
        'gfft_comm_rank': (c.c_int, [vp, ip, ip]),
        'gfft_comm_destroy': (c.c_int, [vp]),
        'gfft_sendrecv': (c.c_int, [vp, c.c_int, c.POINTER(Msg), c.c_int, c.POINTER(Msg), vp]),
        'gfft_alltoallv': (c.c_int, [vp, vp, i64p, i64p, vp, i64p, i64p, c.c_int, vp]),
        'gfft_stream_create': (c.c_int, [c.POINTER(vp)]),
        'gfft_stream_destroy': (c.c_int, [vp]),
        'gfft_stream_wait_event': (c.c_int, [vp, vp]),
        'gfft_event_create_untimed': (c.c_int, [c.POINTER(vp)]),
        'gfft_probe_tile_copy': (c.c_int, [vp, vp, c.c_int64, c.c_int64, c.c_int64, c.c_int, vp]),
    }
    for name, (res, args) in sigs.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    return sigs


EXPORTS = None


def lib():
    """The loaded shared library; raises if it has not been built."""
    global _lib, EXPORTS
    if _lib is None:
        if not os.path.exists(LIBPATH):
            raise ImportError(
                "libgfft.so is not built (%s). Build it with `python -c 'import __graft_entry__ as g; "
                "g.build()'` or `make -C mpi4py-fft_amd/csrc`. There is no CPU fallback." % LIBPATH)
        _lib = ctypes.CDLL(LIBPATH)
        EXPORTS = _declare(_lib)
    return _lib


def check_async():
    """Raise if a launch that already returned has failed since the last look (gfft_async_error: a fused pass pair
    that gave up a wait).  Called where the host is about to READ results, after the synchronisation."""
    if _lib is not None:
        check(_lib.gfft_async_error())


def check(rc):
    if rc != 0:
        l = lib()
        detail = l.gfft_last_error().decode() or l.gfft_exchange_last_error().decode()
        raise GfftError('%s: %s' % (l.gfft_strerror(rc).decode(), detail))


def check_wire(rc):
    """status of an exchange-module call (its error text is kept separately from the planner's)"""
    if rc != 0:
        l = lib()
        raise GfftError('%s: %s' % (l.gfft_strerror(rc).decode(), l.gfft_exchange_last_error().decode()))


def _i64(seq):
    return (ctypes.c_int64 * len(seq))(*[int(s) for s in seq])


_raw_stream = None


def current_stream():
    """hipStream_t of torch's current stream (torch owns device memory and streams here).
    Uses torch's raw-handle accessor when it exists: the public `current_stream()` builds a Stream
    object per call, ~7 us of the ~16 us a small transform costs on the host."""
    global _raw_stream
    import torch
    if _raw_stream is None:
        fast = getattr(torch._C, '_cuda_getCurrentRawStream', None)
        if fast is not None and torch.cuda.is_available():
            _raw_stream = lambda: ctypes.c_void_p(fast(torch.cuda.current_device()))
        elif torch.cuda.is_available():
            _raw_stream = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        else:
            _raw_stream = lambda: ctypes.c_void_p(0)
    return _raw_stream()


def precision_of(dtype):
    return 8 if np.dtype(dtype).char in 'dD' else 4


class HipEngine:
    """The product engine: every method is one call through the C ABI."""
    name = 'hip'

    def require_device(self, tensor):
        if tensor.device.type != 'cuda':
            raise GfftError('array lives on %s: libgfft runs on HIP device memory only '
                            '(no CPU fallback)' % tensor.device)

    def plan_create(self, sizes_in, sizes_out, axes, kind, precision):
        h = ctypes.c_void_p()
        ax = (ctypes.c_int * len(axes))(*[int(a) for a in axes])
        check(lib().gfft_plan_create(ctypes.byref(h), len(sizes_in), _i64(sizes_in), _i64(sizes_out),
                                     len(axes), ax, int(kind), int(precision)))
        return h

    def plan_create_r2r(self, sizes, axes, kinds, precision):
        h = ctypes.c_void_p()
        ax = (ctypes.c_int * len(axes))(*[int(a) for a in axes])
        kd = (ctypes.c_int * len(kinds))(*[int(k) for k in kinds])
        check(lib().gfft_plan_create_r2r(ctypes.byref(h), len(sizes), _i64(sizes), len(axes), ax, kd,
                                         int(precision)))
        return h

    def plan_create_guru(self, precision, kind, dim, howmany, in_blocks=1, in_block_stride=0, out_blocks=1,
                         out_block_stride=0, n_keep=0):
        """Strided batched 1-D plan (gfft_plan_create_guru); dim / howmany entries are (n, is, os).
        n_keep: entries kept on the truncated side of a fused 3/2-rule truncation / zero padding
        (gfft_plan_create_guru_padded).  None when the engine has no single-pass kernel for it."""
        h = ctypes.c_void_p()
        hm = (IoDim * max(1, len(howmany)))(*[IoDim(*[int(x) for x in d]) for d in howmany])
        rc = lib().gfft_plan_create_guru_padded(ctypes.byref(h), int(precision), int(kind), ctypes.byref(IoDim(*[int(x) for x in dim])),
                                                int(n_keep), len(howmany), hm, int(in_blocks), int(in_block_stride),
                                                int(out_blocks), int(out_block_stride))
        if rc == -2:
            return None
        check(rc)
        return h

    def plan_create_guru2(self, precision, kind, cols, rows, planes, cols_first=False, in_blocks=1, in_block_stride=0,
                          out_blocks=1, out_block_stride=0):
        """Batched 2-D plan, plane by plane, as ONE launch where a fused pair exists (gfft_plan_create_guru2): the two
        local stages of a slab-decomposed transform.  cols / rows / planes are (n, is, os).  None when the engine has
        no single-pass kernels for the lengths; `plan_cost(h)[2]` tells whether it runs as one launch or two."""
        h = ctypes.c_void_p()
        io = lambda d: ctypes.byref(IoDim(*[int(x) for x in d]))
        rc = lib().gfft_plan_create_guru2(ctypes.byref(h), int(precision), int(kind), io(cols), io(rows), io(planes),
                                          1 if cols_first else 0, int(in_blocks), int(in_block_stride), int(out_blocks),
                                          int(out_block_stride))
        if rc == -2:
            return None
        check(rc)
        return h

    def execute_ptr(self, h, ptr_in, ptr_out, scale, stream=None):
        """gfft_execute on raw device addresses (element 0 of the plan's input / output)."""
        check(lib().gfft_execute(h, ctypes.c_void_p(ptr_in), ctypes.c_void_p(ptr_out), float(scale),
                                 current_stream() if stream is None else stream))

    def plan_status(self, h):
        """gfft_plan_status: raises if a launch of THIS plan voided itself since anybody last looked (call after a sync)."""
        check(lib().gfft_plan_status(h))

    def plan_execute(self, h, tin, tout, scale):
        self.require_device(tin)
        self.require_device(tout)
        check(lib().gfft_execute(h, tin.data_ptr(), tout.data_ptr(), float(scale), current_stream()))

    def plan_create_padded(self, padded, kept, kind, precision):
        """The padded 3-D transform of a one-rank PFFT as one plan, or None when libgfft keeps the
        per-axis form (gfft_plan_create_padded)."""
        h = ctypes.c_void_p()
        rc = lib().gfft_plan_create_padded(ctypes.byref(h), _i64(padded), _i64(kept), int(kind), int(precision))
        if rc == -2:
            return None
        check(rc)
        return h

    def plan_set_truncation(self, h, n_keep):
        """True if the truncation/padding was fused into the plan, False if it cannot be."""
        rc = lib().gfft_plan_set_truncation(h, int(n_keep))
        if rc == -2:
            return False
        check(rc)
        return True

    def plan_set_split(self, h, side, nblocks):
        """Packed (all-to-all buffer) layout on the plan's input (side 0) / output (side 1);
        True if fused, False if this plan cannot (caller keeps the pack / unpack kernel)."""
        rc = lib().gfft_plan_set_split(h, int(side), int(nblocks))
        if rc == -2:
            return False
        check(rc)
        return True

    def plan_set_tiles(self, h, side, tile, tile_stride):
        """Tile-major exchange-buffer layout on one side of a one-pass plan (gfft_plan_set_tiles);
        False if the plan's kernel cannot address it."""
        rc = lib().gfft_plan_set_tiles(h, int(side), int(tile), int(tile_stride))
        if rc == -2:
            return False
        check(rc)
        return True

    def plan_set_flat(self, h, body_width=0, tail_offset=0, tail_row_stride=0):
        """Tiles over the flattened batch dims of a strided plan, optionally reading rows stored as
        body + leftover columns (gfft_plan_set_flat)."""
        rc = lib().gfft_plan_set_flat(h, int(body_width), int(tail_offset), int(tail_row_stride))
        if rc == -2:
            return False
        check(rc)
        return True

    def plan_set_split_slabs(self, h, side, nblocks, rows_per_slab, tile):
        """gfft_plan_set_split_slabs: uneven blocks of packed-real rows, slab by slab, tile-major."""
        rc = lib().gfft_plan_set_split_slabs(h, int(side), int(nblocks), int(rows_per_slab), int(tile))
        if rc == -2:
            return False
        check(rc)
        return True

    def plan_destroy(self, h):
        if h is not None and _lib is not None:
            _lib.gfft_plan_destroy(h)

    def plan_describe(self, h):
        buf = ctypes.create_string_buffer(4096)
        check(lib().gfft_plan_describe(h, buf, 4096))
        return buf.value.decode()

    def plan_cost(self, h):
        f, b, n = ctypes.c_double(), ctypes.c_double(), ctypes.c_int()
        check(lib().gfft_plan_cost(h, ctypes.byref(f), ctypes.byref(b), ctypes.byref(n)))
        return f.value, b.value, n.value

    def plan_profile(self, h, npasses):
        """[(kernel family, algorithmic bytes per launch, total ms, launches)] per pass since the
        last call; needs set_option('profile', 1) during the executes."""
        ms = (ctypes.c_float * npasses)()
        n = ctypes.c_int()
        check(lib().gfft_plan_profile(h, ms, npasses, ctypes.byref(n)))
        out = []
        for i in range(npasses):
            buf = ctypes.create_string_buffer(64)
            b = ctypes.c_double()
            check(lib().gfft_plan_pass_info(h, i, buf, 64, ctypes.byref(b)))
            out.append((buf.value.decode(), b.value, float(ms[i]), n.value))
        return out

    def pack(self, tarray, tpacked, shape, axis, nparts, itemsize):
        self.require_device(tarray)
        check(lib().gfft_pack(tarray.data_ptr(), tpacked.data_ptr(), len(shape), _i64(shape), axis,
                              nparts, itemsize, current_stream()))

    def unpack(self, tpacked, tarray, shape, axis, nparts, itemsize):
        self.require_device(tarray)
        check(lib().gfft_unpack(tpacked.data_ptr(), tarray.data_ptr(), len(shape), _i64(shape), axis,
                                nparts, itemsize, current_stream()))

    def pack_ptr(self, ptr_array, ptr_packed, shape, axis, nparts, itemsize, unpack=False):
        """gfft_pack / gfft_unpack on raw device addresses (sub-arrays of exchange buffers, pipeline.py)."""
        fn = lib().gfft_unpack if unpack else lib().gfft_pack
        a, b = (ptr_packed, ptr_array) if unpack else (ptr_array, ptr_packed)
        check(fn(ctypes.c_void_p(a), ctypes.c_void_p(b), len(shape), _i64(shape), axis, nparts, itemsize, current_stream()))

    def truncate(self, tpadded, ttrunc, shape_padded, axis, n_trunc, is_real, precision, scale):
        self.require_device(tpadded)
        check(lib().gfft_truncate(tpadded.data_ptr(), ttrunc.data_ptr(), len(shape_padded),
                                  _i64(shape_padded), axis, n_trunc, int(is_real), precision,
                                  float(scale), current_stream()))

    def pad(self, ttrunc, tpadded, shape_padded, axis, n_trunc, is_real, precision):
        self.require_device(tpadded)
        check(lib().gfft_pad(ttrunc.data_ptr(), tpadded.data_ptr(), len(shape_padded),
                             _i64(shape_padded), axis, n_trunc, int(is_real), precision,
                             current_stream()))

    def scale(self, t, count, precision, scale):
        self.require_device(t)
        check(lib().gfft_scale(t.data_ptr(), count, precision, float(scale), current_stream()))

    # pseudo-spectral caller kernels (spectral.py)
    def ps_curl(self, tu, tout, k, shape, precision):
        self.require_device(tu)
        check(lib().gfft_ps_curl(tu.data_ptr(), tout.data_ptr(), k[0].data_ptr(), k[1].data_ptr(), k[2].data_ptr(),
                                 shape[0], shape[1], shape[2], precision, current_stream()))

    def ps_cross(self, ta, tb, tout, count, precision):
        self.require_device(ta)
        check(lib().gfft_ps_cross(ta.data_ptr(), tb.data_ptr(), tout.data_ptr(), count, precision, current_stream()))

    def ps_project(self, tdu, tu, k, shape, nu, precision):
        self.require_device(tdu)
        check(lib().gfft_ps_project(tdu.data_ptr(), tu.data_ptr(), k[0].data_ptr(), k[1].data_ptr(), k[2].data_ptr(),
                                    shape[0], shape[1], shape[2], float(nu), precision, current_stream()))

    def ps_rk_stage(self, tu, tu0, tu1, tdu, count, cb, ca, precision):
        self.require_device(tu1)
        check(lib().gfft_ps_rk_stage(None if tu is None else tu.data_ptr(), None if tu0 is None else tu0.data_ptr(),
                                     tu1.data_ptr(), tdu.data_ptr(), count, float(cb), float(ca), precision,
                                     current_stream()))

    def copy(self, tsrc, tdst):
        self.require_device(tsrc)
        check(lib().gfft_memcpy_d2d(tdst.data_ptr(), tsrc.data_ptr(),
                                    tsrc.numel() * tsrc.element_size(), current_stream()))


_engine = HipEngine()


def engine():
    return _engine


def set_engine(e):
    """TEST SEAM ONLY (see module docstring).  Returns the previous engine."""
    global _engine
    old, _engine = _engine, (e if e is not None else HipEngine())
    return old


def set_option(key, value):
    check(lib().gfft_set_option(key.encode(), int(value)))


def device_count():
    n = ctypes.c_int(0)
    rc = lib().gfft_device_count(ctypes.byref(n))
    return n.value if rc == 0 else 0
