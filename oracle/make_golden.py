#!/usr/bin/env python3
"""Golden-fixture generator: runs the REFERENCE's own Python (mpi4py-fft, /root/reference)
and stores small input/output vectors under tests/golden/.

TEST INFRASTRUCTURE ONLY.  Runs in the build container only (needs /root/reference); the
fixtures it writes are data (shapes, index maps, arrays) and are what travels to the GPU box.

How the reference is made importable here (nothing from it is copied into the repo):
  * a temporary directory receives a copy of /root/reference/mpi4py_fft (deleted on exit);
  * mpi4py_fft/fftw/utilities.pyx (the reference's own Cython source) is compiled there with
    the container's cython + gcc, so `mpi4py_fft.fftw.aligned` etc. are the reference's code;
  * `mpi4py` is absent from this image, so an in-process emulation of the handful of MPI calls
    the path touches is injected as `mpi4py.MPI`: every virtual rank is
    a Python thread, `Alltoallw` really exchanges the subarray blocks between the threads.
    The reference's mpifft.py / pencil.py / libfft.py / distarray.py run unchanged on top,
    with backend='numpy' (libfftw3 is not installed; fftw/factory.py:24-41 guards that import).

Usage:  python oracle/make_golden.py            # rewrites tests/golden/*.npz
"""
import os
import sys
import shutil
import subprocess
import tempfile
import threading
import types
import itertools

import numpy as np

REF = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), 'tests', 'golden')


# --------------------------------------------------------------------------------------
# In-process MPI emulation (threads = ranks)
# --------------------------------------------------------------------------------------
class _Datatype:
    def __init__(self, char):
        self.char = char
        self.slices = None

    def Create_subarray(self, sizes, subsizes, starts):
        t = _Datatype(self.char)
        t.sizes = tuple(sizes)
        t.slices = tuple(slice(s, s + n) for s, n in zip(starts, subsizes))
        return t

    def Commit(self):
        return self

    def Free(self):
        pass


class _World:
    """State shared by the threads of one emulated mpiexec run."""
    def __init__(self, size):
        self.size = size
        self.lock = threading.Lock()
        self.groups = {}

    def group(self, key, members):
        with self.lock:
            g = self.groups.get(key)
            if g is None:
                g = types.SimpleNamespace(members=tuple(members),
                                          barrier=threading.Barrier(len(members)),
                                          box={})
                self.groups[key] = g
            return g


_tls = threading.local()
_uid = itertools.count()


class Comm:
    """Emulated intra-communicator: `members` are world ranks in communicator-rank order."""
    def __init__(self, world, members, me, key, topo=None):
        self._world = world
        self._members = tuple(members)
        self._me = me                      # world rank of the calling thread
        self._key = key
        self._topo = topo                  # None or (dims, coords-of-me)
        self._g = world.group(key, members)

    # -- queries
    def Get_size(self):
        return len(self._members)

    def Get_rank(self):
        return self._members.index(self._me)

    size = property(Get_size)
    rank = property(Get_rank)

    def Is_inter(self):
        return False

    def Get_topology(self):
        return MPI.CART if self._topo is not None else MPI.UNDEFINED

    def Get_dim(self):
        return len(self._topo[0])

    def __eq__(self, other):
        return isinstance(other, Comm) and self._key == other._key

    def __ne__(self, other):
        return not self.__eq__(other)

    def __hash__(self):
        return hash(self._key)

    def __bool__(self):
        return True

    def Free(self):
        pass

    # -- topology (row-major Cartesian grid, sub-rank = coordinate)
    def Create_cart(self, dims, periods=None, reorder=False):
        dims = tuple(int(d) for d in dims)
        assert int(np.prod(dims)) == self.Get_size()
        coords = np.unravel_index(self.Get_rank(), dims)
        return Comm(self._world, self._members, self._me,
                    ('cart', self._key, dims), topo=(dims, tuple(int(c) for c in coords)))

    def Sub(self, remdims):
        dims, mycoords = self._topo
        remdims = tuple(bool(r) for r in remdims)
        members = []
        for r, wr in enumerate(self._members):
            c = np.unravel_index(r, dims)
            if all(remdims[i] or c[i] == mycoords[i] for i in range(len(dims))):
                members.append(wr)
        fixed = tuple(None if remdims[i] else mycoords[i] for i in range(len(dims)))
        subdims = tuple(d for d, r in zip(dims, remdims) if r)
        subcoords = tuple(c for c, r in zip(mycoords, remdims) if r)
        return Comm(self._world, members, self._me, ('sub', self._key, remdims, fixed),
                    topo=(subdims, subcoords))

    # -- collectives
    def _exchange(self, value):
        g = self._g
        g.box[self._me] = value
        g.barrier.wait()
        vals = [g.box[m] for m in g.members]
        g.barrier.wait()
        return vals

    def bcast(self, obj, root=0):
        return self._exchange(obj)[root]

    def gather(self, obj, root=0):
        vals = self._exchange(obj)
        return vals if self.Get_rank() == root else None

    def allreduce(self, obj, op=None):
        return sum(self._exchange(obj))

    def reduce(self, obj, op=None, root=0):
        v = sum(self._exchange(obj))
        return v if self.Get_rank() == root else None

    def barrier(self):
        self._exchange(None)

    Barrier = barrier

    def Alltoallw(self, sendspec, recvspec):
        """MPI_Alltoallw with counts=1, displs=0 and subarray datatypes
        (the only form pencil.py:182-183,200-201 uses)."""
        sbuf, _, stypes = sendspec
        rbuf, _, rtypes = recvspec
        me = self.Get_rank()
        # what I send to peer i, already "packed" (row-major copy of the subarray)
        out = [np.ascontiguousarray(sbuf[t.slices]) for t in stypes]
        allout = self._exchange(out)
        for j, t in enumerate(rtypes):
            rbuf[t.slices] = allout[j][me].reshape(rbuf[t.slices].shape)


class _WorldProxy:
    """MPI.COMM_WORLD: resolves to the calling thread's world communicator."""
    def __getattr__(self, name):
        return getattr(_tls.world_comm, name)

    def __eq__(self, other):
        return _tls.world_comm == other

    def __hash__(self):
        return hash(_tls.world_comm)


def _compute_dims(nnodes, dims):
    """MPI_Dims_create as MPICH does it: balanced factors, non-increasing, zeros are free."""
    dims = [int(d) for d in (dims if np.ndim(dims) else [0] * int(dims))]
    fixed = int(np.prod([d for d in dims if d > 0])) if any(d > 0 for d in dims) else 1
    assert nnodes % fixed == 0
    rem = nnodes // fixed
    nfree = sum(1 for d in dims if d == 0)
    if nfree == 0:
        return dims
    # prime factors, largest first, each assigned to the currently smallest bin
    f, n, p = [], rem, 2
    while n > 1:
        while n % p == 0:
            f.append(p)
            n //= p
        p += 1
    bins = [1] * nfree
    for q in sorted(f, reverse=True):
        bins[bins.index(min(bins))] *= q
    bins.sort(reverse=True)
    it = iter(bins)
    return [d if d > 0 else next(it) for d in dims]


MPI = types.ModuleType('mpi4py.MPI')
MPI.CART = 1
MPI.UNDEFINED = -32766
MPI.Comm = Comm
MPI.Intracomm = Comm
MPI.COMM_WORLD = _WorldProxy()
MPI.Compute_dims = _compute_dims
MPI._typedict = {c: _Datatype(c) for c in 'fdgFDGbBhHiIlLqQ'}
MPI.SUM = 'sum'


def _self_comm():
    w = _World(1)
    return Comm(w, (0,), 0, ('self', next(_uid)))


class _SelfProxy:
    def __getattr__(self, name):
        if not hasattr(_tls, 'self_comm'):
            _tls.self_comm = _self_comm()
        return getattr(_tls.self_comm, name)

    def __eq__(self, other):
        return isinstance(other, (Comm, _SelfProxy)) and other.Get_size() == 1 and \
            getattr(other, '_key', ('self',))[0] == 'self'

    def __hash__(self):
        return 1


MPI.COMM_SELF = _SelfProxy()


def mpirun(nranks, fn):
    """Run fn(comm_world) on `nranks` emulated ranks; returns the list of results."""
    world = _World(nranks)
    results = [None] * nranks
    errors = []

    def target(r):
        try:
            _tls.world_comm = Comm(world, tuple(range(nranks)), r, ('world', id(world)))
            results[r] = fn(_tls.world_comm)
        except BaseException as e:  # noqa
            import traceback
            errors.append((r, traceback.format_exc()))
            for g in list(world.groups.values()):
                g.barrier.abort()

    threads = [threading.Thread(target=target, args=(r,)) for r in range(nranks)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errors:
        raise RuntimeError('emulated rank failed:\n' + errors[0][1])
    return results


# --------------------------------------------------------------------------------------
# Import the reference
# --------------------------------------------------------------------------------------
def import_reference(tmp):
    pkg = os.path.join(tmp, 'mpi4py_fft')
    shutil.copytree(os.path.join(REF, 'mpi4py_fft'), pkg)
    # build the reference's own utilities.pyx (aligned(), FFTW_* constants)
    import sysconfig
    fdir = os.path.join(pkg, 'fftw')
    subprocess.check_call([sys.executable, '-m', 'cython', '-3', 'utilities.pyx'], cwd=fdir)
    ext = sysconfig.get_config_var('EXT_SUFFIX')
    subprocess.check_call(['gcc', '-O1', '-shared', '-fPIC', '-w',
                           '-I', sysconfig.get_paths()['include'], '-I', np.get_include(),
                           'utilities.c', '-o', 'utilities' + ext], cwd=fdir)
    mpi4py = types.ModuleType('mpi4py')
    mpi4py.MPI = MPI
    sys.modules['mpi4py'] = mpi4py
    sys.modules['mpi4py.MPI'] = MPI
    sys.path.insert(0, tmp)
    _tls.world_comm = Comm(_World(1), (0,), 0, ('world', 'import'))
    import mpi4py_fft  # noqa  (the reference)
    return mpi4py_fft


def rng_array(shape, dtype, seed):
    """The synthetic-input rule shared with tests/ and bench.py."""
    rng = np.random.default_rng(seed)
    dtype = np.dtype(dtype)
    a = rng.standard_normal(shape)
    if dtype.kind == 'c':
        a = a + 1j * rng.standard_normal(shape)
    return a.astype(dtype)


# --------------------------------------------------------------------------------------
# Fixture families
# --------------------------------------------------------------------------------------
def gen_blockdist(ref):
    from mpi4py_fft.pencil import _blockdist
    rows = []
    for N in (1, 2, 7, 8, 9, 12, 13, 64, 513, 1024, 1025, 2048):
        for p in (1, 2, 3, 4, 8):
            if N < p:
                continue
            for r in range(p):
                n, s = _blockdist(N, p, r)
                rows.append((N, p, r, n, s))
    return dict(blockdist=np.array(rows, dtype=np.int64))


PFFT_CASES = [
    # name, P, shape, dtype, kwargs
    ('c2c_8x6x4_p1', 1, (8, 6, 4), 'D', {}),
    ('c2c_16x12x10_p1', 1, (16, 12, 10), 'D', {}),
    ('c2c_16x12x10_p1_collapse', 1, (16, 12, 10), 'D', dict(collapse=True)),
    ('c2c_7x8x9_p1', 1, (7, 8, 9), 'D', {}),
    ('c2c_12x13_p1', 1, (12, 13), 'D', {}),
    ('c2c_32c_p1_f32', 1, (32, 32, 32), 'F', {}),
    ('r2c_16x12x10_p1', 1, (16, 12, 10), 'd', {}),
    ('r2c_7x8x9_p1', 1, (7, 8, 9), 'd', {}),
    ('r2c_8x6x4_axes201_p1', 1, (8, 6, 4), 'd', dict(axes=(2, 0, 1))),
    ('r2c_16x12x10_p1_pad', 1, (16, 12, 10), 'd', dict(padding=[1.5, 1.5, 1.5])),
    ('c2c_16x12x10_p1_pad', 1, (16, 12, 10), 'D', dict(padding=[1.5, 1.5, 1.5])),
    ('r2c_12x13_p1_f32', 1, (12, 13), 'f', {}),
    ('c2c_16x12x10_p2', 2, (16, 12, 10), 'D', {}),
    ('c2c_7x8x9_p2', 2, (7, 8, 9), 'D', {}),
    ('r2c_16x12x10_p2', 2, (16, 12, 10), 'd', {}),
    ('c2c_16x12x10_p4', 4, (16, 12, 10), 'D', {}),
    ('c2c_16x12x10_p4_slab', 4, (16, 12, 10), 'D', dict(grid=(-1,))),
    ('r2c_13x12x10_p4', 4, (13, 12, 10), 'd', {}),
    ('r2c_16x12x10_p4_pad', 4, (16, 12, 10), 'd', dict(padding=[1.5, 1.5, 1.5])),
    ('c2c_16x16x16_p8', 8, (16, 16, 16), 'D', {}),
    ('r2c_16x16x18_p8', 8, (16, 16, 18), 'd', {}),
    ('c2c_12x13_p2', 2, (12, 13), 'D', {}),
    ('c2c_6x7x8x9_p4', 4, (6, 7, 8, 9), 'D', dict(axes=((0,), (1,), (2, 3)))),
]


def gen_pfft(ref):
    """PFFT forward / backward of the reference on emulated ranks: geometry + values."""
    from mpi4py_fft import PFFT, newDistArray
    out = {}
    for name, P, shape, dt, kw in PFFT_CASES:
        gin = rng_array(shape if not kw.get('padding') else
                        tuple(int(np.floor(n * p)) for n, p in zip(shape, kw['padding'])),
                        dt, seed=1234)

        def run(comm, shape=shape, dt=dt, kw=kw, gin=gin):
            kw = {k: (list(v) if isinstance(v, list) else v) for k, v in kw.items()}
            fft = PFFT(comm, shape, dtype=dt, backend='numpy', **kw)
            u = newDistArray(fft, False)
            u[...] = gin[fft.local_slice(False)]
            uh = fft.forward(u).copy()
            ub = fft.backward(uh.copy()).copy()
            pin, pout = fft.pencil
            info = dict(
                axes=[list(a) for a in fft.axes],
                grid=[c.Get_size() for c in fft.subcomm],
                in_subshape=pin.subshape, in_substart=pin.substart, in_axis=pin.axis,
                out_subshape=pout.subshape, out_substart=pout.substart, out_axis=pout.axis,
                gshape_in=fft.global_shape(False), gshape_out=fft.global_shape(True),
                nxfftn=len(fft.xfftn), ntransfer=len(fft.transfer),
                transfers=[(t.comm.Get_size(), t.shape, t.subshapeA, t.axisA, t.subshapeB, t.axisB)
                           for t in fft.transfer],
            )
            return info, np.asarray(uh), np.asarray(ub)

        res = mpirun(P, run)
        out[name + '/input'] = gin
        out[name + '/P'] = np.int64(P)
        out[name + '/dtype'] = np.array(dt)
        out[name + '/shape'] = np.array(shape, dtype=np.int64)
        import json
        out[name + '/kw'] = np.array(json.dumps(kw))
        for r, (info, uh, ub) in enumerate(res):
            out['%s/r%d/info' % (name, r)] = np.array(json.dumps(
                {k: (list(map(lambda x: list(x) if isinstance(x, tuple) else x, v))
                     if isinstance(v, (list, tuple)) else v) for k, v in info.items()},
                default=lambda o: int(o)))
            out['%s/r%d/fwd' % (name, r)] = uh
            out['%s/r%d/bwd' % (name, r)] = ub
    return out


def gen_transfer(ref):
    """Pencil.transfer block maps: tests/test_pencil.py's loop on emulated ranks, but storing
    what each rank holds after trans1.forward / trans2.forward."""
    from mpi4py_fft.pencil import Subcomm, Pencil
    out = {}
    cases = [(2, (7, 8, 9), None, (0, 1, 2)), (4, (7, 8, 9), None, (2, 1, 0)),
             (4, (8, 9, 7), 1, (1, 2, 0)), (2, (8, 9), None, (0, 1, 0)),
             (3, (7, 9, 8), None, (1, 0, 2)), (8, (9, 8, 16), None, (2, 0, 1))]
    for ci, (P, shape, pdim, (a1, a2, a3)) in enumerate(cases):
        G = np.arange(int(np.prod(shape)), dtype='d').reshape(shape)

        def run(comm, shape=shape, pdim=pdim, a1=a1, a2=a2, a3=a3, G=G):
            subcomm = Subcomm(comm, pdim)
            p0 = Pencil(subcomm, shape)
            pA = p0.pencil(a1)
            pB = pA.pencil(a2)
            pC = pB.pencil(a3)
            t1 = Pencil.transfer(pA, pB, 'd')
            t2 = Pencil.transfer(pB, pC, 'd')
            sl = tuple(slice(s, s + n) for s, n in zip(pA.substart, pA.subshape))
            A = np.ascontiguousarray(G[sl])
            B = np.zeros(pB.subshape)
            C = np.zeros(pC.subshape)
            t1.forward(A, B)
            t2.forward(B, C)
            A2 = np.zeros_like(A)
            B2 = np.zeros_like(B)
            t2.backward(C, B2)
            t1.backward(B2, A2)
            assert np.array_equal(A2, A)
            geo = np.array([pA.subshape, pA.substart, pB.subshape, pB.substart,
                            pC.subshape, pC.substart], dtype=np.int64)
            return geo, A, B, C, [c.Get_size() for c in subcomm]

        res = mpirun(P, run)
        key = 'transfer%d' % ci
        out[key + '/P'] = np.int64(P)
        out[key + '/shape'] = np.array(shape, dtype=np.int64)
        out[key + '/axes'] = np.array([a1, a2, a3], dtype=np.int64)
        out[key + '/pdim'] = np.int64(-1 if pdim is None else pdim)
        for r, (geo, A, B, C, sizes) in enumerate(res):
            out['%s/r%d/geo' % (key, r)] = geo
            out['%s/r%d/A' % (key, r)] = A
            out['%s/r%d/B' % (key, r)] = B
            out['%s/r%d/C' % (key, r)] = C
            out['%s/r%d/grid' % (key, r)] = np.array(sizes, dtype=np.int64)
    return out


def gen_geometry(ref):
    """Appendix A: per-rank geometry of the BASELINE configs from the reference's Pencil."""
    from mpi4py_fft import PFFT
    out = {}
    # PFFT planning allocates U,V per group with numpy backend -> only small stand-ins run
    # end to end; the big configs use Pencil arithmetic alone.
    from mpi4py_fft.pencil import Subcomm, Pencil
    cfgs = [('C3', 2, (512, 512, 512), False), ('C4', 1, (1024,) * 3, False),
            ('C4', 2, (1024,) * 3, False), ('C4', 4, (1024,) * 3, False),
            ('C4', 8, (1024,) * 3, False), ('C5', 8, (2048,) * 3, True)]
    for name, P, shape, real in cfgs:
        def run(comm, shape=shape, real=real):
            sub = Subcomm(comm, [0, 0, 1])
            p0 = Pencil(sub, shape, 2)
            shp = list(shape)
            if real:
                shp[2] = shp[2] // 2 + 1
            pA = Pencil(sub, shp, 2)
            pB = pA.pencil(1)
            pC = pB.pencil(0)
            return np.array([p0.subshape, p0.substart, pA.subshape, pA.substart,
                             pB.subshape, pB.substart, pC.subshape, pC.substart,
                             [c.Get_size() for c in sub]], dtype=np.int64)
        res = mpirun(P, run)
        out['%s_P%d' % (name, P)] = np.stack(res)
    return out


def gen_libfft(ref):
    """libfft.FFT (numpy backend): serial multi-axis forward with the reference's scaling,
    incl. padding/truncation (libfft.py:263-311)."""
    from mpi4py_fft.libfft import FFT
    out = {}
    cases = [('D', (8,), None, False), ('D', (7,), None, False), ('d', (8,), None, False),
             ('d', (9,), None, False), ('D', (8, 9), (0,), False), ('D', (8, 9), (1, 0), False),
             ('d', (8, 9), (0, 1), False), ('d', (7, 8, 9), (1,), False),
             ('d', (7, 8, 9), (2, 0), False), ('D', (7, 8, 9), None, False),
             ('d', (12, 9), (0,), 1.5), ('D', (12, 9), (1,), 1.5), ('D', (13, 9), (0,), 1.5),
             ('d', (8, 12), (1,), 1.5), ('d', (8, 16), (1,), 2.0), ('D', (16, 8), (0,), 2.0),
             ('F', (16, 12), None, False), ('f', (16, 12), None, False)]
    for i, (dt, shape, axes, padding) in enumerate(cases):
        fft = FFT(shape, axes, dtype=dt, backend='numpy', padding=padding)
        A = rng_array(shape, dt, seed=77 + i)
        B = fft.forward(A.copy()).copy()
        A2 = fft.backward(B.copy()).copy()
        k = 'libfft%d' % i
        out[k + '/dtype'] = np.array(dt)
        out[k + '/shape'] = np.array(shape, dtype=np.int64)
        out[k + '/axes'] = np.array([-99] if axes is None else list(axes), dtype=np.int64)
        out[k + '/padding'] = np.float64(0 if padding is False else padding)
        out[k + '/A'] = A
        out[k + '/B'] = B
        out[k + '/A2'] = A2
    # docstring known-answer tests (xfftn.py:85-88, 220-223, 293-301)
    out['kat/fftn_in'] = np.array([1, 2, 3, 4], dtype='D')
    out['kat/fftn_out'] = np.array([10, -2 + 2j, -2, -2 - 2j], dtype='D')
    out['kat/rfftn_in'] = np.array([1, 2, 3, 4], dtype='d')
    out['kat/rfftn_out'] = np.array([10, -2 + 2j, -2], dtype='D')
    out['kat/irfftn_in'] = np.array([1, 2, 3, 4], dtype='D')
    out['kat/irfftn_out6'] = np.array([15., -4., 0., -1., 0., -4.])
    out['kat/irfftn_out7'] = np.array([19., -5.04891734, -0.30797853, -0.64310413,
                                       -0.64310413, -0.30797853, -5.04891734])
    return out


def main():
    os.makedirs(OUT, exist_ok=True)
    tmp = tempfile.mkdtemp(prefix='refimport_')
    try:
        ref = import_reference(tmp)
        assert ref.__version__ == '2.0.6'
        for name, fn in (('blockdist', gen_blockdist), ('geometry', gen_geometry),
                         ('transfer', gen_transfer), ('libfft', gen_libfft),
                         ('pfft', gen_pfft)):
            data = fn(ref)
            path = os.path.join(OUT, name + '.npz')
            np.savez_compressed(path, **data)
            print('wrote %s (%d entries, %.1f KiB)' % (path, len(data), os.path.getsize(path) / 1024))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == '__main__':
    main()
