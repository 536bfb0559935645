"""CHECKER engine for CPU-only tests of the host logic (TEST INFRASTRUCTURE).

Implements the `_lib.HipEngine` interface with numpy on host tensors so that pencil arithmetic,
exchange plans, buffer chaining and the communicator layer can be exercised where no GPU exists
(the `-m "not gpu"` suite, incl. the world_size-2 gloo tests).  It is injected explicitly with
`_lib.set_engine(HostEngine())` by tests; the package never selects it.  The arithmetic is the
oracle's (numpy pocketfft), so a test using this engine checks orchestration, not kernels.
"""
import numpy as np
import torch

from oracle import pfft_oracle as O


def _np(t):
    return t.numpy()


class HostEngine:
    name = 'host-checker'

    def require_device(self, tensor):
        pass

    def plan_create(self, sizes_in, sizes_out, axes, kind, precision):
        return dict(sizes_in=tuple(sizes_in), sizes_out=tuple(sizes_out), axes=tuple(axes),
                    kind=kind, precision=precision)

    def plan_create_r2r(self, sizes, axes, kinds, precision):
        return dict(sizes_in=tuple(sizes), sizes_out=tuple(sizes), axes=tuple(axes), kind='r2r',
                    kinds=tuple(kinds), precision=precision)

    def plan_set_split(self, h, side, nblocks):
        # same acceptance rule as gfft_plan_set_split for what this checker can see
        n = h['sizes_in'][h['axes'][0]]
        if len(h['axes']) != 1 or h['kind'] not in (-1, 1) or h['kind'] == 'r2r' or nblocks & (nblocks - 1) or nblocks > 8 or n % nblocks:
            return False
        h['split_in' if side == 0 else 'split_out'] = nblocks
        return True

    def plan_execute(self, h, tin, tout, scale):
        if h.get('split_in', 1) > 1 or h.get('split_out', 1) > 1:
            # packed layouts: unpack the input / pack the output around the natural transform
            nat = dict(h, split_in=1, split_out=1)
            tin2, tout2 = tin, tout
            if h.get('split_in', 1) > 1:
                tin2 = torch.empty_like(tin)
                self.unpack(tin, tin2, h['sizes_in'], h['axes'][0], h['split_in'], 0)
            if h.get('split_out', 1) > 1:
                tout2 = torch.empty_like(tout)
            self.plan_execute(nat, tin2, tout2, scale)
            if h.get('split_out', 1) > 1:
                self.pack(tout2, tout, h['sizes_out'], h['axes'][0], h['split_out'], 0)
            return
        a = _np(tin).reshape(h['sizes_in'])
        axes, kind = h['axes'], h['kind']
        if kind == 'r2r':
            r = a
            for ax, k in zip(axes, h['kinds']):
                r = O.r2r_1d(r, ax, k)
            _np(tout).reshape(h['sizes_out'])[...] = (r * scale).astype(_np(tout).dtype)
            return
        if kind == -1:
            r = np.fft.fftn(a, axes=axes)
        elif kind == 1:
            r = np.fft.ifftn(a, axes=axes) * np.prod([a.shape[i] for i in axes])
        elif kind == -2:
            r = np.fft.rfftn(a, axes=axes)
        else:
            s = [h['sizes_out'][i] for i in axes]
            r = np.fft.irfftn(a, s=s, axes=axes) * np.prod(s)
        _np(tout).reshape(h['sizes_out'])[...] = (r * scale).astype(_np(tout).dtype)

    # strided batched 1-D plans on raw addresses (gfft_plan_create_guru / gfft_execute): what the
    # chunked pipeline (pipeline.py) drives; host memory here, so that its layouts, chunk offsets and
    # exchange plans can be checked over gloo without a GPU
    def plan_create_guru(self, precision, kind, dim, howmany, in_blocks=1, in_block_stride=0, out_blocks=1,
                         out_block_stride=0):
        n = int(dim[0])
        if n & (n - 1) or n < 16 or len(howmany) > 3:          # acceptance of the register kernels, roughly
            return None
        for nb in (in_blocks, out_blocks):
            if nb & (nb - 1) or nb > 8 or n % nb:
                return None
        return dict(guru=True, precision=precision, kind=kind, dim=tuple(int(x) for x in dim),
                    howmany=[tuple(int(x) for x in d) for d in howmany], inb=(in_blocks, in_block_stride),
                    outb=(out_blocks, out_block_stride))

    def execute_ptr(self, h, ptr_in, ptr_out, scale, stream=None):
        import ctypes
        n, es_in, es_out = h['dim']
        cdt = np.complex128 if h['precision'] == 8 else np.complex64
        isz = np.dtype(cdt).itemsize

        def view(ptr, es, blocks, which):
            nb, bs = blocks
            per = n // nb
            shape = [d[0] for d in h['howmany']] + [nb, per]
            strides = [d[which] for d in h['howmany']] + [bs if nb > 1 else per * es, es]
            extent = 1 + sum((sz - 1) * abs(st) for sz, st in zip(shape, strides))
            flat = np.frombuffer((ctypes.c_char * (extent * isz)).from_address(ptr), dtype=cdt)
            return np.lib.stride_tricks.as_strided(flat, shape=shape, strides=[st * isz for st in strides])
        vin = view(ptr_in, es_in, h['inb'], 1)
        vout = view(ptr_out, es_out, h['outb'], 2)
        lines = np.ascontiguousarray(vin).reshape(vin.shape[:-2] + (n,))
        r = np.fft.fft(lines, axis=-1) if h['kind'] == -1 else np.fft.ifft(lines, axis=-1) * n
        vout[...] = (r * scale).astype(cdt).reshape(vout.shape)

    def plan_set_truncation(self, h, n_keep):
        return False

    def plan_destroy(self, h):
        pass

    def plan_describe(self, h):
        return 'host checker plan %r' % (h,)

    def plan_cost(self, h):
        return 0.0, 0.0, 0

    def pack(self, tarray, tpacked, shape, axis, nparts, itemsize):
        a = _np(tarray).reshape(shape)
        out = _np(tpacked).reshape(-1)
        pos = 0
        for i in range(nparts):
            n, s = O.blockdist(shape[axis], nparts, i)
            ix = [slice(None)] * len(shape)
            ix[axis] = slice(s, s + n)
            blk = np.ascontiguousarray(a[tuple(ix)]).reshape(-1)
            out[pos:pos + blk.size] = blk
            pos += blk.size

    def unpack(self, tpacked, tarray, shape, axis, nparts, itemsize):
        a = _np(tarray).reshape(shape)
        src = _np(tpacked).reshape(-1)
        pos = 0
        for i in range(nparts):
            n, s = O.blockdist(shape[axis], nparts, i)
            ix = [slice(None)] * len(shape)
            ix[axis] = slice(s, s + n)
            sz = a[tuple(ix)].size
            a[tuple(ix)] = src[pos:pos + sz].reshape(a[tuple(ix)].shape)
            pos += sz

    def _offt(self, shape_padded, axis, n_trunc, is_real):
        # an OFFT whose truncation helpers we borrow: fake a padded transform along `axis`
        f = O.OFFT.__new__(O.OFFT)
        f.axes = (axis,)
        f.real = bool(is_real)
        full = list(shape_padded)
        out = list(shape_padded)
        out[axis] = n_trunc
        f.full_out_shape, f.out_shape = tuple(full), tuple(out)
        return f

    def truncate(self, tpadded, ttrunc, shape_padded, axis, n_trunc, is_real, precision, scale):
        f = self._offt(shape_padded, axis, n_trunc, is_real)
        r = f._truncate(_np(tpadded).reshape(shape_padded)) * scale
        _np(ttrunc).reshape(f.out_shape)[...] = r

    def pad(self, ttrunc, tpadded, shape_padded, axis, n_trunc, is_real, precision):
        f = self._offt(shape_padded, axis, n_trunc, is_real)
        _np(tpadded).reshape(shape_padded)[...] = f._pad(_np(ttrunc).reshape(f.out_shape))

    def scale(self, t, count, precision, scale):
        t.mul_(scale)

    def copy(self, tsrc, tdst):
        tdst.copy_(tsrc)
