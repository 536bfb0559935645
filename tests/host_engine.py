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
                         out_block_stride=0, n_keep=0):
        n = int(dim[0])
        if n_keep and n_keep != n:                              # (fused truncation: GPU engine only)
            return None
        if n & (n - 1) or n < 16 or len(howmany) > 3:          # acceptance of the register kernels, roughly
            return None
        for nb in (in_blocks, out_blocks):
            if nb & (nb - 1) or nb > 8 or n % nb:
                return None
        return dict(guru=True, precision=precision, kind=kind, dim=tuple(int(x) for x in dim),
                    howmany=[tuple(int(x) for x in d) for d in howmany], inb=(in_blocks, in_block_stride),
                    outb=(out_blocks, out_block_stride))

    def plan_create_guru2(self, precision, kind, cols, rows, planes, cols_first=False, in_blocks=1, in_block_stride=0,
                          out_blocks=1, out_block_stride=0):
        """gfft_plan_create_guru2: batched 2-D transform plane by plane, the strided axis in blocks on one side"""
        n1, n2 = int(cols[0]), int(rows[0])
        if n1 & (n1 - 1) or n2 & (n2 - 1) or n1 < 16 or n2 < 16 or tuple(rows[1:]) != (1, 1):
            return None
        if in_blocks > 1 and out_blocks > 1:
            return None
        for nb in (in_blocks, out_blocks):
            if nb & (nb - 1) or nb > 8 or n1 % nb:
                return None
        return dict(guru2=True, precision=precision, kind=kind, cols=tuple(int(x) for x in cols), n2=n2,
                    planes=tuple(int(x) for x in planes), inb=(in_blocks, in_block_stride), outb=(out_blocks, out_block_stride))

    def _execute_guru2(self, h, ptr_in, ptr_out, scale):
        import ctypes
        cdt = np.complex128 if h['precision'] == 8 else np.complex64
        isz = np.dtype(cdt).itemsize
        n1, c_i, c_o = h['cols']
        n2 = h['n2']
        npl, p_i, p_o = h['planes']

        def offsets(which):
            es, ps = (c_i, p_i) if which == 0 else (c_o, p_o)
            nb, bs = h['inb'] if which == 0 else h['outb']
            e = np.arange(n1)
            per = n1 // nb
            line = (e // per) * bs + (e % per) * es if nb > 1 else e * es
            return np.arange(npl)[:, None, None] * ps + line[None, :, None] + np.arange(n2)[None, None, :]
        oi, oo = offsets(0), offsets(1)
        fin = np.frombuffer((ctypes.c_char * ((int(oi.max()) + 1) * isz)).from_address(ptr_in), dtype=cdt)
        fout = np.frombuffer((ctypes.c_char * ((int(oo.max()) + 1) * isz)).from_address(ptr_out), dtype=cdt)
        a = fin[oi]
        r = np.fft.fft2(a, axes=(1, 2)) if h['kind'] == -1 else np.fft.ifft2(a, axes=(1, 2)) * (n1 * n2)
        fout[oo] = (r * scale).astype(cdt)

    def plan_set_tiles(self, h, side, tile, tile_stride):
        """gfft_plan_set_tiles: the transformed axis (plans along a contiguous axis) or the adjacent
        columns (strided plans) of one side are tile-major."""
        if not h.get('guru') or tile & (tile - 1):
            return False
        n, es_in, es_out = h['dim']
        rows = es_in == 1 and es_out == 1
        nb = (h['inb'] if side == 0 else h['outb'])[0]
        if rows and tile and ((n // nb) % tile or n < 128):
            return False
        h.setdefault('tiles', {})[side] = (int(tile), int(tile_stride)) if tile else None
        return True

    def plan_set_flat(self, h, body_width=0, tail_offset=0, tail_row_stride=0):
        if not h.get('guru'):
            return False
        h['flat'] = (int(body_width), int(tail_offset), int(tail_row_stride))
        return True

    def execute_ptr(self, h, ptr_in, ptr_out, scale, stream=None):
        import ctypes
        if h.get('guru2'):
            return self._execute_guru2(h, ptr_in, ptr_out, scale)
        n, es_in, es_out = h['dim']
        cdt = np.complex128 if h['precision'] == 8 else np.complex64
        isz = np.dtype(cdt).itemsize
        rows = es_in == 1 and es_out == 1
        hm = list(h['howmany'])
        while len(hm) < 3:
            hm.insert(0, (1, 0, 0))
        (no, o_i, o_o), (nm, m_i, m_o), (ni, i_i, i_o) = hm
        flat = h.get('flat')

        def offsets(which):
            """element offsets [o][m][i][e] of one side"""
            es = es_in if which == 0 else es_out
            nb, bs = h['inb'] if which == 0 else h['outb']
            so, sm, si = (o_i, m_i, i_i) if which == 0 else (o_o, m_o, i_o)
            tl = (h.get('tiles') or {}).get(which)
            e = np.arange(n)
            per = n // nb
            eb, ew = e // per, e % per
            if rows and tl:
                line = eb * (bs if nb > 1 else 0) + (ew // tl[0]) * tl[1] + ew % tl[0]
                if nb == 1:
                    line = (e // tl[0]) * tl[1] + e % tl[0]
            else:
                line = eb * bs + ew * es if nb > 1 else e * es
            i = np.arange(ni)
            col = ((i // tl[0]) * tl[1] + i % tl[0]) if (tl and not rows) else i * si
            off = (np.arange(no)[:, None, None, None] * so + np.arange(nm)[None, :, None, None] * sm
                   + col[None, None, :, None] + line[None, None, None, :])
            if which == 0 and flat and flat[0]:
                bw, toff, tms = flat
                tail = (np.arange(no)[:, None, None, None] * so + toff + np.arange(nm)[None, :, None, None] * tms
                        + (i - bw)[None, None, :, None] + line[None, None, None, :])
                off = np.where((i >= bw)[None, None, :, None], tail, off)
            return off
        oi, oo = offsets(0), offsets(1)
        fin = np.frombuffer((ctypes.c_char * ((int(oi.max()) + 1) * isz)).from_address(ptr_in), dtype=cdt)
        fout = np.frombuffer((ctypes.c_char * ((int(oo.max()) + 1) * isz)).from_address(ptr_out), dtype=cdt)
        lines = fin[oi]
        r = np.fft.fft(lines, axis=-1) if h['kind'] == -1 else np.fft.ifft(lines, axis=-1) * n
        fout[oo] = (r * scale).astype(cdt)

    def plan_set_truncation(self, h, n_keep):
        return False

    def plan_destroy(self, h):
        pass

    def plan_describe(self, h):
        return 'host checker plan %r' % (h,)

    def plan_cost(self, h):
        return 0.0, 0.0, (1 if h.get('guru2') else 0)

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
