"""CPU oracle for the PFFT hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A numpy restatement of the algorithm mpi4py-fft runs for `PFFT.forward/backward`
(/root/reference/mpi4py_fft).  Only tests/, __graft_entry__.smoke() and bench.py's
`cpu_baseline` leg may import this module; the shipped package never does.

Parity status: PINNED.  tests/test_oracle_golden.py checks every function here against
fixtures (tests/golden/*.npz) produced by running the reference's own Python in the build
container (oracle/make_golden.py), and against the reference's docstring known-answer vectors
(fftw/xfftn.py:85-88, 220-223, 293-301).

All ranks of a run are simulated in ONE process: a distributed array is a list of per-rank
numpy arrays.  The serial engine is numpy's pocketfft (what the reference's backend='numpy'
uses, libfft.py:81-102); libfftw3 -- the engine behind backend='fftw' -- is a third-party
dependency that is neither vendored in the reference nor installed in this image (unpinned:
setup.py:64-81 links whatever libfftw3 it finds), so arithmetic parity is against the DFT
definition FFTW documents (unnormalised, exp(-2 pi i jk/n) forward), which pocketfft also
implements.
"""
import itertools
import numpy as np

try:  # used only to thread the cpu_baseline; numerics are identical to numpy.fft
    import scipy.fft as _sfft
except Exception:  # pragma: no cover
    _sfft = None


# ---------------------------------------------------------------- decomposition arithmetic
def blockdist(N, size, rank):
    """Length and start of `rank`'s block of an axis of length N split over `size` ranks.
    Restates pencil.py:5-9 (`_blockdist`)."""
    base, extra = divmod(int(N), int(size))
    return base + (rank < extra), rank * base + min(rank, extra)


def compute_dims(nranks, dims):
    """MPI_Dims_create (MPICH): fill the zero entries of `dims` with a balanced,
    non-increasing factorisation of nranks / prod(nonzero dims).  (pencil.py:79)"""
    dims = [max(0, int(d)) for d in dims]
    fixed = int(np.prod([d for d in dims if d > 0], dtype=np.int64)) if any(dims) else 1
    assert nranks % fixed == 0, "grid does not divide communicator size"
    rest, primes, p = nranks // fixed, [], 2
    while rest > 1:
        while rest % p == 0:
            primes.append(p)
            rest //= p
        p += 1
    free = [1] * dims.count(0)
    for q in sorted(primes, reverse=True):
        free[free.index(min(free))] *= q
    free = iter(sorted(free, reverse=True))
    return [d if d > 0 else next(free) for d in dims]


def rank_coords(rank, dims):
    """Row-major Cartesian coordinates; the sub-communicator rank along grid axis i is
    coords[i] (MPI_Cart_create + MPI_Cart_sub, pencil.py:80-88)."""
    return tuple(int(c) for c in np.unravel_index(rank, dims))


class OPencil:
    """Geometry of one rank's block.  `sizes[i]`, `ranks[i]` = size of / rank in the
    sub-communicator that axis i is distributed over.  Restates pencil.py:277-323."""
    def __init__(self, sizes, ranks, shape, axis):
        nd = len(shape)
        axis %= nd
        assert nd >= 2 and min(shape) >= 1 and len(sizes) == nd and sizes[axis] == 1
        self.sizes, self.ranks = tuple(sizes), tuple(ranks)
        self.shape, self.axis = tuple(int(s) for s in shape), axis
        ns = [blockdist(shape[i], sizes[i], ranks[i]) for i in range(nd)]
        for i in range(nd):
            assert shape[i] >= sizes[i]
        self.subshape = tuple(n for n, _ in ns)
        self.substart = tuple(s for _, s in ns)

    def pencil(self, axis):
        """Re-align on `axis`: the two axes swap their sub-communicators (pencil.py:320-323)."""
        axis %= len(self.shape)
        sizes, ranks = list(self.sizes), list(self.ranks)
        i, j = self.axis, axis
        sizes[i], sizes[j] = sizes[j], sizes[i]
        ranks[i], ranks[j] = ranks[j], ranks[i]
        return OPencil(sizes, ranks, self.shape, axis)

    def local_slice(self):
        return tuple(slice(s, s + n) for s, n in zip(self.substart, self.subshape))


def transfer_groups(dims, grid_axis):
    """World ranks grouped into the sub-communicators of Cartesian grid axis `grid_axis`;
    position in each list = rank in the sub-communicator."""
    groups = {}
    for r in range(int(np.prod(dims))):
        c = rank_coords(r, dims)
        groups.setdefault(c[:grid_axis] + c[grid_axis + 1:], []).append(r)
    return list(groups.values())


def transfer(arrays, group, axisA, axisB, shapeB_axis_global):
    """Global redistribution inside one sub-communicator `group` (list of indices into
    `arrays`): input blocks aligned on axisA (full along it, split along axisB), output
    aligned on axisB.  Explicit pack -> exchange -> unpack of what pencil.py:12-29,168-183
    moves with MPI_Alltoallw + subarray datatypes.  Returns the new per-member arrays."""
    p = len(group)
    A = [arrays[g] for g in group]
    NA = A[0].shape[axisA]
    outs = []
    for me in range(p):
        nA, sA = blockdist(NA, p, me)
        shp = list(A[me].shape)
        shp[axisA] = nA
        shp[axisB] = shapeB_axis_global
        B = np.empty(shp, dtype=A[me].dtype)
        for j in range(p):            # block received from peer j
            nB, sB = blockdist(shapeB_axis_global, p, j)
            assert A[j].shape[axisB] == nB
            src = [slice(None)] * B.ndim
            src[axisA] = slice(sA, sA + nA)
            packed = np.ascontiguousarray(A[j][tuple(src)])        # pack (sender side)
            dst = [slice(None)] * B.ndim
            dst[axisB] = slice(sB, sB + nB)
            B[tuple(dst)] = packed.reshape(B[tuple(dst)].shape)    # unpack (receiver side)
        outs.append(B)
    return outs


# ---------------------------------------------------------------- serial transforms
def _fftmod(workers):
    if workers and workers > 1 and _sfft is not None:
        return _sfft, dict(workers=workers)
    return np.fft, {}


# FFTW's real-to-real kinds (utilities.pyx:12-19) in scipy's (family, type) numbering; scipy's
# unnormalised ("backward") transforms are FFTW's definitions -- the reference's own tests pin its
# r2r plans against scipy.fftpack the same way (tests/test_fftw.py:101-118).
R2R_KINDS = {3: ('dct', 1), 5: ('dct', 2), 4: ('dct', 3), 6: ('dct', 4),
             7: ('dst', 1), 9: ('dst', 2), 8: ('dst', 3), 10: ('dst', 4)}
R2R_INVERSE = {3: 3, 4: 5, 5: 4, 6: 6, 7: 7, 8: 9, 9: 8, 10: 10}       # xfftn.py:818-826


def r2r_1d(a, axis, kind):
    import scipy.fft
    fam, typ = R2R_KINDS[int(kind)]
    f = scipy.fft.dct if fam == 'dct' else scipy.fft.dst
    return f(np.asarray(a, dtype=np.result_type(a.dtype, np.float32)), type=typ, axis=axis, norm=None)


def r2r_logical_n(kind, n):
    """Per-axis normalisation length (xfftn.py:763-816)."""
    return 2 * (n - 1) if kind == 3 else 2 * (n + 1) if kind == 7 else 2 * n


class OFFT:
    """Serial transform over `axes` of an array of `shape` with the reference's conventions
    (libfft.py:376-422): forward multiplies by 1/prod(N_axes) unless normalize=False, backward
    is unscaled unless normalize=True; real input -> r2c along axes[-1]; optional 3/2-rule
    padding on a single axis (libfft.py:263-311)."""
    def __init__(self, shape, axes=None, dtype='d', padding=False, workers=None, r2r=None):
        self.shape = tuple(int(s) for s in shape)
        nd = len(self.shape)
        self.axes = tuple(range(nd)) if axes is None else tuple(a % nd for a in np.atleast_1d(axes))
        self.dtype = np.dtype(dtype)
        assert self.dtype.char in 'fdFD'
        self.real = self.dtype.kind == 'f'
        self.cdtype = np.dtype(self.dtype.char.upper())
        self.M = 1.0 / float(np.prod([self.shape[a] for a in self.axes]))
        self.workers = workers
        # r2r: FFTW kind (3..10) of the FORWARD transform, one int for all axes of the group or one
        # per axis (the `transforms=` dict of libfft.py:330-340 with dctn / dstn planners)
        self.r2r = None
        if r2r is not None:
            assert self.real and padding is False
            self.r2r = [int(r2r)] * len(self.axes) if np.ndim(r2r) == 0 else [int(k) for k in r2r]
            self.M = 1.0 / float(np.prod([r2r_logical_n(k, self.shape[a]) for k, a in zip(self.r2r, self.axes)]))
            self.cdtype = self.dtype
            self.padding_factor = 1.0
            self.full_out_shape = self.out_shape = self.shape
            return
        pf = padding[self.axes[-1]] if np.ndim(padding) else (padding or 1.0)
        self.padding_factor = float(pf) if padding is not False else 1.0
        out = list(self.shape)
        if self.real:
            out[self.axes[-1]] = out[self.axes[-1]] // 2 + 1
        self.full_out_shape = tuple(out)
        if abs(self.padding_factor - 1.0) > 1e-8:
            assert len(self.axes) == 1
            ax = self.axes[0]
            n = int(np.round(self.shape[ax] / self.padding_factor))
            out[ax] = n // 2 + 1 if self.real else n
        self.out_shape = tuple(out)

    @property
    def padded(self):
        return abs(self.padding_factor - 1.0) > 1e-8

    def _truncate(self, Vfull):
        ax = self.axes[-1]
        N = self.out_shape[ax]
        N0 = N      # libfft.py:267: `self.forward.output_array` IS the truncated array
        ix = [slice(None)] * Vfull.ndim
        if self.real:
            ix[ax] = slice(0, N)
            T = Vfull[tuple(ix)].copy()
            if N0 % 2 == 0:
                ix[ax] = N - 1
                T[tuple(ix)] = 2 * T[tuple(ix)].real
            return T
        T = np.zeros(self.out_shape, dtype=Vfull.dtype)
        ix[ax] = slice(0, N // 2 + 1)
        T[tuple(ix)] = Vfull[tuple(ix)]
        if N // 2 > 0:
            ix[ax] = slice(-(N // 2), None)
            T[tuple(ix)] += Vfull[tuple(ix)]
        return T

    def _pad(self, T):
        ax = self.axes[-1]
        N = self.out_shape[ax]
        N0 = N      # libfft.py:290 (same remark)
        V = np.zeros(self.full_out_shape, dtype=T.dtype)
        ix = [slice(None)] * T.ndim
        if self.real:
            ix[ax] = slice(0, N)
            V[tuple(ix)] = T
            if N0 % 2 == 0:
                ix[ax] = N - 1
                V[tuple(ix)] = 0.5 * V[tuple(ix)].real
            return V
        ix[ax] = slice(0, N // 2 + 1)
        V[tuple(ix)] = T[tuple(ix)]
        if N // 2 > 0:
            ix[ax] = slice(-(N // 2), None)
            V[tuple(ix)] = T[tuple(ix)]
        if N0 % 2 == 0:
            for k in (N // 2, -(N // 2)):
                ix[ax] = k
                V[tuple(ix)] *= 0.5
        return V

    def forward(self, u, normalize=True):
        mod, kw = _fftmod(self.workers)
        u = np.asarray(u)
        assert u.shape == self.shape
        if self.r2r is not None:
            V = u
            for a, k in zip(self.axes, self.r2r):
                V = r2r_1d(V, a, k)
            return (V * self.M if normalize else V).astype(self.dtype, copy=False)
        s = [self.shape[a] for a in self.axes]
        V = (mod.rfftn if self.real else mod.fftn)(u, s=s, axes=self.axes, **kw)
        V = V.astype(self.cdtype, copy=False)
        if self.padded:
            V = self._truncate(V)
        if normalize:
            V = V * V.real.dtype.type(self.M) if V.dtype == np.complex64 else V * self.M
        return V.astype(self.cdtype, copy=False)

    def backward(self, V, normalize=False):
        mod, kw = _fftmod(self.workers)
        V = np.asarray(V)
        assert V.shape == self.out_shape, (V.shape, self.out_shape)
        if self.r2r is not None:
            u = V
            for a, k in zip(self.axes, self.r2r):
                u = r2r_1d(u, a, R2R_INVERSE[k])
            return (u * self.M if normalize else u).astype(self.dtype, copy=False)
        if self.padded:
            V = self._pad(V)
        s = [self.shape[a] for a in self.axes]
        if self.real:
            u = mod.irfftn(V, s=s, axes=self.axes, **kw)
        else:
            u = mod.ifftn(V, s=s, axes=self.axes, **kw)
        u = u * (1.0 / self.M)        # numpy scales its inverse by 1/N; the reference does not
        if normalize:
            u = u * self.M
        return u.astype(self.dtype, copy=False)


# ---------------------------------------------------------------- the parallel transform
def normalize_axes(axes, ndim):
    """axes argument -> list of tuples of non-negative ints (mpifft.py:213-240)."""
    if axes is None:
        axes = list(range(ndim))
    elif isinstance(axes, (int, np.integer)):
        axes = [int(axes)]
    groups = []
    for g in axes:
        g = (int(g),) if isinstance(g, (int, np.integer)) else tuple(int(a) for a in g)
        g = tuple(a % ndim if a < 0 else a for a in g)
        assert len(g) > 0 and len(set(g)) == len(g) and max(g) < ndim
        groups.append(g)
    return groups


class OPFFT:
    """All ranks of a PFFT in one process.  Restates mpifft.py:202-347 (planning) and
    mpifft.py:46-79 (execution): FFT_0, T_0, FFT_1, T_1, ... with the group order
    axes[-1], axes[-2], ... forward and the mirror backward."""
    def __init__(self, nranks, shape, axes=None, dtype='d', grid=None, padding=False,
                 collapse=False, workers=None, r2r=None):
        # r2r: {axes group (tuple): forward FFTW kind(s)} -- the oracle's spelling of `transforms=`
        r2r = {} if r2r is None else {tuple(k): v for k, v in r2r.items()}
        shape = [int(s) for s in shape]
        nd = len(shape)
        groups = normalize_axes(axes, nd)
        dtype = np.dtype(dtype)
        if padding is not False:
            padding = list(padding)
            assert len(padding) == nd
            for g in groups:
                if len(g) == 1 and padding[g[0]] > 1.0 + 1e-6:
                    old = float(shape[g[0]])
                    shape[g[0]] = int(np.floor(shape[g[0]] * padding[g[0]]))
                    padding[g[0]] = shape[g[0]] / old
        self.input_shape = tuple(shape)
        if grid is not None:
            dims = list(grid) + [1] * (nd - len(grid))
        else:
            dims = [0] * nd
            for a in groups[-1]:
                dims[a] = 1
        self.dims = compute_dims(nranks, dims)
        assert all(self.dims[a] == 1 for a in groups[-1])
        if collapse:
            merged = [[]]
            for g in reversed(groups):
                if all(self.dims[a] == 1 for a in g):
                    merged[0] = list(g) + merged[0]
                else:
                    merged.insert(0, list(g))
            groups = [tuple(g) for g in merged]
        self.axes = tuple(tuple(g) for g in groups)
        self.nranks = nranks
        self.coords = [rank_coords(r, self.dims) for r in range(nranks)]

        # per rank: pencils, serial transforms; shared: transfer descriptions
        self.ffts = [[] for _ in range(nranks)]       # [rank][stage]
        self.transfers = []                           # (axisA, axisB, grid_axis_of_comm, N_B)
        self.pencil_in, self.pencil_out = [], []
        for r in range(nranks):
            shp = list(shape)
            dt = dtype
            sizes = list(self.dims)
            ranks = list(self.coords[r])
            # which grid axis each array axis is currently distributed over
            owner = list(range(nd))
            g = self.axes[-1]
            pen = OPencil(sizes, ranks, shp, g[-1])
            self.pencil_in.append(pen)
            f = OFFT(pen.subshape, g, dt, padding, workers, r2r.get(tuple(g)))
            self.ffts[r].append(f)
            if f.out_shape[g[-1]] != shp[g[-1]]:
                dt = f.cdtype
                shp[g[-1]] = f.out_shape[g[-1]]
                pen = OPencil(pen.sizes, pen.ranks, shp, g[-1])
            trs = []
            for g in reversed(self.axes[:-1]):
                penB = pen.pencil(g[-1])
                # communicator = the one axis g[-1] was distributed over before the swap
                trs.append((pen.axis, penB.axis, owner[g[-1]], shp[g[-1]]))
                owner[pen.axis], owner[penB.axis] = owner[penB.axis], owner[pen.axis]
                f = OFFT(penB.subshape, g, dt, padding, workers, r2r.get(tuple(g)))
                self.ffts[r].append(f)
                pen = penB
                if f.out_shape[g[-1]] != shp[g[-1]]:
                    dt = f.cdtype
                    shp[g[-1]] = f.out_shape[g[-1]]
                    pen = OPencil(penB.sizes, penB.ranks, shp, g[-1])
            self.pencil_out.append(pen)
            if r == 0:
                self.transfers = trs
                self.output_shape = tuple(shp)
                self.dtype_in, self.dtype_out = dtype, np.dtype(dt)

    def scatter(self, G, forward_output=False):
        pens = self.pencil_out if forward_output else self.pencil_in
        return [np.ascontiguousarray(G[p.local_slice()]) for p in pens]

    def gather(self, arrays, forward_output=True):
        pens = self.pencil_out if forward_output else self.pencil_in
        G = np.zeros(pens[0].shape, dtype=arrays[0].dtype)
        for p, a in zip(pens, arrays):
            G[p.local_slice()] = a
        return G

    def _do_transfer(self, arrays, t, backward=False):
        axisA, axisB, gaxis, NB = t
        out = [None] * self.nranks
        for group in transfer_groups(self.dims, gaxis):
            if backward:
                NA = sum(arrays[g].shape[axisA] for g in group)
                res = transfer(arrays, group, axisB, axisA, NA)
            else:
                res = transfer(arrays, group, axisA, axisB, NB)
            for g, a in zip(group, res):
                out[g] = a
        return out

    def forward(self, arrays, normalize=True):
        cur = list(arrays)
        nst = len(self.transfers)
        for i in range(nst):
            cur = [self.ffts[r][i].forward(cur[r], normalize) for r in range(self.nranks)]
            cur = self._do_transfer(cur, self.transfers[i])
        return [self.ffts[r][nst].forward(cur[r], normalize) for r in range(self.nranks)]

    def backward(self, arrays, normalize=False):
        cur = list(arrays)
        nst = len(self.transfers)
        for i in range(nst, 0, -1):
            cur = [self.ffts[r][i].backward(cur[r], normalize) for r in range(self.nranks)]
            cur = self._do_transfer(cur, self.transfers[i - 1], backward=True)
        return [self.ffts[r][0].backward(cur[r], normalize) for r in range(self.nranks)]


def rng_array(shape, dtype, seed):
    """Synthetic-input rule shared by fixtures, tests and bench.py:
    default_rng(seed).standard_normal (+ 1j * standard_normal), C-contiguous."""
    rng = np.random.default_rng(seed)
    dtype = np.dtype(dtype)
    a = rng.standard_normal(shape)
    if dtype.kind == 'c':
        a = a + 1j * rng.standard_normal(shape)
    return a.astype(dtype)
