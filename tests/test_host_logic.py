"""CPU tests of the host logic (planning, pencils, exchange plans, buffer chaining) with the
checker engine injected -- NOT a product path (see tests/host_engine.py)."""
import numpy as np
import pytest

from tests import cases
from tests.host_engine import HostEngine


@pytest.fixture(autouse=True)
def host_engine():
    from mpi4py_fft_amd import _lib
    old = _lib.set_engine(HostEngine())
    yield
    _lib.set_engine(old)


@pytest.mark.parametrize('name', cases.pfft_case_names())
def test_pfft_matches_reference_fixture(name):
    cases.check_pfft_golden(name)


@pytest.mark.parametrize('ci', range(6))
def test_transfer_matches_reference_fixture(ci):
    cases.check_transfer_golden(ci)


@pytest.mark.parametrize('P,shape,dt,kw', [
    (1, (12, 13), 'd', dict(axes=(-1, 0))),
    (2, (12, 13, 5), 'D', dict(axes=((0,), (1, 2)))),
    (4, (12, 13, 12, 13), 'd', dict(axes=((0,), (1,), (2, 3)))),
    (4, (13, 12, 12), 'F', dict(grid=(-1,), collapse=True)),
    (3, (9, 8, 7), 'd', {}),
])
def test_pfft_vs_oracle(P, shape, dt, kw):
    cases.check_pfft_vs_oracle(P, shape, dt, **kw)


def test_random_redistribute_chains():
    """Random DistArray redistribution walks (see tests/test_gpu_random.py) on the host logic."""
    rng = np.random.default_rng(77)
    done = 0
    while done < 25:
        done += bool(cases.check_redistribute_chain(rng))


def test_distarray_api():
    from mpi4py_fft_amd import DistArray, newDistArray, PFFT, comm
    from tests import thread_comm

    def body(c):
        # tests/test_darray.py: properties, tensors, redistribute conserves the norm
        from mpi4py_fft_amd import Subcomm
        N = (8, 10, 12)
        sub = Subcomm(c, [0, 0, 1])
        z = DistArray(N, subcomm=sub, dtype=float, alignment=2)
        z[...] = np.random.default_rng(c.Get_rank()).random(z.shape)
        assert z.global_shape == N and z.dimensions == 3 and z.rank == 0
        assert z.alignment == 2 and z.commsizes[2] == 1
        n0 = c.allgather_obj(float(np.sum(np.asarray(z) ** 2)))
        z1 = z.redistribute(1)
        assert z1.alignment == 1 and z1.global_shape == N
        n1 = c.allgather_obj(float(np.sum(np.asarray(z1) ** 2)))
        assert np.isclose(sum(n0), sum(n1))
        z2 = z1.redistribute(out=DistArray(N, subcomm=sub, dtype=float, alignment=2))
        assert np.allclose(np.asarray(z2), np.asarray(z))
        v = DistArray((3,) + N, subcomm=sub, dtype=float, alignment=2, rank=1)
        v[...] = 1.0
        w = v.redistribute(0)
        assert w.rank == 1 and w.shape[0] == 3 and w.alignment == 0
        assert np.all(np.asarray(w) == 1.0)
        fft = PFFT(c, darray=z1)
        assert fft.forward.input_array.shape == z1.shape
        uh = newDistArray(fft, True)
        assert uh.dtype == np.dtype('D') and uh.shape == fft.forward.output_array.shape
        return True

    assert all(thread_comm.run(4, body))


def test_misuse_raises():
    from mpi4py_fft_amd import PFFT, comm, FFT
    with pytest.raises(AssertionError):
        PFFT(comm.COMM_SELF, (8, 8), axes=(0, 0))
    with pytest.raises(AssertionError):
        PFFT(comm.COMM_SELF, (8, 8), dtype='i')
    with pytest.raises(NotImplementedError):
        FFT((8, 8), dtype='D', backend='numpy')
    from mpi4py_fft_amd import fftw
    with pytest.raises(NotImplementedError):
        fftw.get_normalization([fftw.FFTW_R2HC], (8,), (0,))


def test_chunked_transfer_pipeline(monkeypatch):
    """The slab-chunked asynchronous exchange (pencil.Transfer._move_chunked) gives the same
    result as the single exchange; forced on by lowering the size threshold."""
    from mpi4py_fft_amd import pencil
    monkeypatch.setattr(pencil.Transfer, 'CHUNK_MIN_BYTES', 0)
    monkeypatch.setattr(pencil.Transfer, 'CHUNKS', 3)
    for ci in range(6):
        cases.check_transfer_golden(ci)
    cases.check_pfft_golden('c2c_16x16x16_p8')
    cases.check_pfft_golden('r2c_16x16x18_p8')
    cases.check_pfft_golden('r2c_13x12x10_p4')


def test_relay_schedule_is_consistent():
    """relay.Schedule: for every pair of ranks the sends of one side match the receives of the
    other in number, order and length, and every scalar of every block is delivered once."""
    from mpi4py_fft_amd import relay
    from mpi4py_fft_amd.pencil import _blockdist
    for W, p, rest, n in [(8, 2, 5, 37), (8, 4, 3, 1000), (4, 2, 7, 129), (6, 3, 1, 5), (6, 2, 11, 64)]:
        groups = [tuple(range(g * p, (g + 1) * p)) for g in range(W // p)]
        meta = []
        for a in range(W):
            mem = groups[a // p]
            ia = mem.index(a)
            # rank a holds `rest + ia` rows of a length-n axis and sends each member its block of it
            meta.append((mem, [(rest + ia) * _blockdist(n, p, i)[0] for i in range(p)]))
        scheds = [relay.Schedule(meta, me) for me in range(W)]
        for rnd in ('r1', 'r2'):
            for x in range(W):
                for y in range(W):
                    s = [m[2] for m in getattr(scheds[x], rnd + '_send') if m[3] == y]
                    r = [m[2] for m in getattr(scheds[y], rnd + '_recv') if m[3] == x]
                    assert s == r, (W, p, rnd, x, y)
                    assert x != y or not s
        for j in range(W):
            mem = groups[j // p]
            total = sum(meta[a][1][mem.index(j)] for a in mem)
            cover = np.zeros(total, int)
            sc = scheds[j]
            if sc.self_copy:
                cover[sc.self_copy[1]:sc.self_copy[1] + sc.self_copy[2]] += 1
            for b, o, ln, peer in sc.r1_recv + sc.r2_recv:
                if b == 'recv':
                    cover[o:o + ln] += 1
            assert (cover == 1).all(), (W, p, j)


@pytest.mark.parametrize('name', ['c2c_16x16x16_p8', 'r2c_16x16x18_p8', 'r2c_13x12x10_p4'])
def test_relayed_exchange(monkeypatch, name):
    """The two-round multi-path exchange (relay.py) delivers what the direct all-to-all does."""
    monkeypatch.setenv('GFFT_RELAY', '1')
    if name not in cases.pfft_case_names():
        pytest.skip('fixture absent')
    from mpi4py_fft_amd import relay
    calls, run = [], relay.Schedule.run
    monkeypatch.setattr(relay.Schedule, 'run', lambda self, *a: (calls.append(1), run(self, *a))[1])
    cases.check_pfft_golden(name)
    assert calls


def test_relayed_transfer_uneven(monkeypatch):
    monkeypatch.setenv('GFFT_RELAY', '1')
    from mpi4py_fft_amd import relay
    monkeypatch.setattr(relay, 'PIECE_GRAIN', 4)
    cases.check_pfft_vs_oracle(4, (12, 13, 12, 13), 'd', axes=((0,), (1,), (2, 3)))
    cases.check_pfft_vs_oracle(6, (13, 11, 9), 'D')
    cases.check_pfft_vs_oracle(8, (17, 16, 9), 'd')


def test_measured_route_choice(monkeypatch):
    """GFFT_RELAY=measure: the first exchange times both routes, all ranks agree on one, and the
    data is right either way."""
    monkeypatch.setenv('GFFT_RELAY', 'measure')
    from mpi4py_fft_amd import pencil
    monkeypatch.setattr(pencil.Transfer, 'RELAY_MIN_BYTES', 0)
    seen, measure = [], pencil.Transfer._measure_routes
    monkeypatch.setattr(pencil.Transfer, '_measure_routes',
                        lambda self, *a: (lambda r: (seen.append((self.comm.Get_size(), r)), r)[1])(measure(self, *a)))
    cases.check_pfft_golden('c2c_16x16x16_p8')
    cases.check_pfft_vs_oracle(4, (20, 12, 16), 'd')
    assert seen and all(r in ('direct', 'relay') for _, r in seen)


@pytest.mark.parametrize('W,p,ratio', [(8, 2, 0.25), (4, 2, 0.5), (8, 4, 0.6), (6, 3, 4 / 7)])
def test_relay_balances_links(W, p, ratio):
    """Per round no directed link carries more than f/2 of a block (relay.py docstring), i.e. the
    two rounds together move f blocks' worth of wire time where the direct exchange moves 1."""
    from mpi4py_fft_amd import relay
    block = 1 << 20
    groups = [tuple(range(g * p, (g + 1) * p)) for g in range(W // p)]
    meta = [(groups[a // p], [block] * p) for a in range(W)]
    assert abs(relay.direct_fraction(p, W) - ratio) < 1e-12
    for rnd in ('r1_send', 'r2_send'):
        load = np.zeros((W, W))
        for me in range(W):
            for b, o, ln, peer in getattr(relay.Schedule(meta, me), rnd):
                load[me, peer] += ln
        assert load.max() <= 1.02 * ratio / 2 * block, (rnd, load.max() / block)


def test_pack_fusion_bookkeeping(monkeypatch):
    """PFFT._fuse_packs: sides whose neighbouring stage can address the exchange buffer are marked
    packed and the transform still matches the oracle; GFFT_FUSE_PACK=0 switches it off."""
    from tests import thread_comm
    from mpi4py_fft_amd import PFFT
    from oracle import pfft_oracle as O

    def body(comm):
        fft = PFFT(comm, (16, 16, 16), dtype='D')
        flags = [(t.packedA, t.packedB) for t in fft.transfer]
        fft2 = PFFT(comm, (16, 12, 16), dtype='d')          # r2c stage 0, axis 1 = 12 over 2 ranks
        flags2 = [(t.packedA, t.packedB) for t in fft2.transfer]
        return flags, flags2
    flags, flags2 = thread_comm.run(4, body)[0]
    assert flags == [(True, True), (True, True)]

    def slab(comm):
        fft = PFFT(comm, (16, 16, 16), dtype='D', grid=(-1,))
        # the in-place axis-1 stage got its own output array so that it can write the send buffer
        return [(t.comm.Get_size(), t.packedA, t.packedB) for t in fft.transfer], \
            fft.xfftn[1].forward.input_array.data_ptr != fft.xfftn[1].forward.output_array.data_ptr
    fl, unshared = thread_comm.run(2, slab)[0]
    assert fl == [(1, False, False), (2, True, True)] and unshared
    cases.check_pfft_vs_oracle(2, (16, 16, 16), 'D', grid=(-1,))
    # r2c output (9 wide) cannot be cut evenly, 12 is no power-of-two block count problem (12 / 2 = 6 fits)
    assert flags2[0][0] is False
    cases.check_pfft_vs_oracle(4, (16, 16, 16), 'D')
    cases.check_pfft_vs_oracle(8, (32, 16, 16), 'F')
    cases.check_pfft_vs_oracle(4, (16, 12, 16), 'd')
    monkeypatch.setenv('GFFT_FUSE_PACK', '0')

    def body0(comm):
        return [(t.packedA, t.packedB) for t in PFFT(comm, (16, 16, 16), dtype='D').transfer]
    assert thread_comm.run(4, body0)[0] == [(False, False), (False, False)]


def test_callers_output_array_is_written_directly():
    """forward(u, out) / backward(uh, out) with device arrays of the planned layout: the last kernel
    writes `out` itself (no copy out of the planned array); other kinds of `out` are copied into."""
    from tests import thread_comm
    from mpi4py_fft_amd import PFFT, newDistArray
    from oracle import pfft_oracle as O

    for P in (1, 2, 4):
        ref = O.OPFFT(P, (12, 8, 10), dtype='d')
        G = O.rng_array((12, 8, 10), 'd', 3)
        want = ref.forward(ref.scatter(G))

        def body(comm):
            fft = PFFT(comm, (12, 8, 10), dtype='d')
            u = newDistArray(fft, False)
            u[...] = G[fft.local_slice(False)]
            out = newDistArray(fft, True)
            planned = fft.forward.output_array
            planned.fill(7)
            r = fft.forward(u, out)
            assert r is out
            assert np.all(np.asarray(planned) == 7)          # untouched: nothing was staged through it
            back = newDistArray(fft, False)
            r2 = fft.backward(out, back)
            assert r2 is back
            host = np.zeros(fft.shape(True), dtype='D')
            fft.forward(u, host)                             # numpy output: copied into
            return np.asarray(out).copy(), np.asarray(back).copy(), host, fft.local_slice(False)
        for r, (uh, back, host, sl) in enumerate(thread_comm.run(P, body)):
            assert np.abs(uh - want[r]).max() < 1e-13
            assert np.abs(host - want[r]).max() < 1e-13
            assert np.abs(back - G[sl]).max() < 1e-13


def test_r2r_transforms_orchestration():
    """`transforms=` with real-to-real planners through PFFT on thread ranks (host checker engine):
    stage dtypes, normalisation (logical lengths 2N / 2(N-1) / 2(N+1)) and the Fourier stage on
    the remaining axis (tests/test_mpifft.py:35-57)."""
    import functools
    from tests import thread_comm
    from mpi4py_fft_amd import PFFT, newDistArray, fftw
    from oracle import pfft_oracle as O
    assert fftw.get_normalization([fftw.FFTW_REDFT00, fftw.FFTW_RODFT00, fftw.FFTW_REDFT10, fftw.FFTW_FORWARD],
                                  (5, 6, 7, 8), (0, 1, 2, 3)) == 1.0 / (8 * 14 * 14 * 8)
    dct = functools.partial(fftw.dctn, type=1)
    idct = functools.partial(fftw.idctn, type=1)
    dst = functools.partial(fftw.dstn, type=4)
    idst = functools.partial(fftw.idstn, type=4)
    shape, axes = (8, 9, 10), ((0,), (1,), (2,))
    tr = {(2,): (dct, idct), (1,): (dst, idst)}
    for P in (1, 2, 4):
        ref = O.OPFFT(P, shape, axes=axes, dtype='d', r2r={(2,): fftw.FFTW_REDFT00, (1,): fftw.FFTW_RODFT11})
        G = O.rng_array(shape, 'd', 4)
        want = ref.forward(ref.scatter(G))

        def body(comm):
            fft = PFFT(comm, shape, axes=axes, dtype='d', transforms=tr)
            u = newDistArray(fft, False)
            u[...] = G[fft.local_slice(False)]
            uh = np.asarray(fft.forward(u)).copy()
            return uh, np.asarray(fft.backward()).copy(), fft.local_slice(False)
        for r, (uh, back, sl) in enumerate(thread_comm.run(P, body)):
            assert uh.dtype == want[r].dtype and np.abs(uh - want[r]).max() < 1e-13
            assert np.abs(back - G[sl]).max() < 1e-12


def test_distarray_get_global_slice():
    """DistArray.get(gslice): the docstring example of distarray.py:196-212 (4 ranks, N = 6^3,
    alignment 0, z[:] = rank -> z.get((0, :, 0)) == [0 0 0 2 2 2] on rank 0), and a 2-D slice."""
    from tests import thread_comm
    from mpi4py_fft_amd import DistArray, Subcomm

    def body(comm):
        N = (6, 6, 6)
        z = DistArray(N, subcomm=Subcomm(comm, [1, 0, 0]), dtype=float, alignment=0)
        z[...] = float(comm.Get_rank())
        g = z.get((0, slice(None), 0))
        G = np.arange(216, dtype=float).reshape(N)
        z[...] = G[z.local_slice()]
        plane = z.get((slice(None), 3, slice(None)))
        return g, plane
    res = thread_comm.run(4, body)
    assert np.array_equal(res[0][0], [0, 0, 0, 2, 2, 2])
    assert np.array_equal(res[0][1], np.arange(216, dtype=float).reshape(6, 6, 6)[:, 3, :])
    assert all(r[0] is None and r[1] is None for r in res[1:])


def test_distarray_get_on_tensor_fields():
    """get(gslice) with leading tensor indices, rank 0-2, 2-D and 3-D (what tests/test_darray.py:30-45,
    82-97 probe): a line of a unit-filled field sums to its length on rank 0."""
    from tests import thread_comm
    from mpi4py_fft_amd import DistArray, Subcomm

    def body(comm):
        got = []
        for N, grid in (((8, 12), [0, 1]), ((8, 6, 10), [0, 0, 1])):
            for rank in (0, 1, 2):
                a = DistArray((len(N),) * rank + N, subcomm=Subcomm(comm, grid), val=1, rank=rank)
                assert a.rank == rank and a.global_shape == (len(N),) * rank + N
                lines = []
                for ax in range(len(N)):
                    g = [0] * len(N)
                    g[ax] = slice(None)
                    lines.append(a.get((0,) * rank + tuple(g)))
                got.append((N, lines))
        return got
    res = thread_comm.run(4, body)
    for N, lines in res[0]:
        for ax, k in enumerate(lines):
            assert len(k) == N[ax] and np.sum(k) == N[ax]
    assert all(k is None for r in res[1:] for _, lines in r for k in lines)


def test_newdistarray_variants_plan_transforms():
    """newDistArray(view / rank / forward_output) and PFFT(darray=component) for every variant
    (the combinations tests/test_darray.py:111-133 walks through)."""
    from tests import thread_comm
    from mpi4py_fft_amd import PFFT, DistArray, newDistArray
    from mpi4py_fft_amd.array import DeviceArray

    def body(comm):
        pfft = PFFT(comm, (8, 8, 8))
        for spectral in (True, False):
            for rank in (0, 1, 2):
                a = newDistArray(pfft, forward_output=spectral, rank=rank)
                assert isinstance(a, DistArray) and a.rank == rank
                comp = a[(0,) * rank] if rank else a
                assert isinstance(comp, DistArray) and comp.rank == 0
                q = PFFT(comm, darray=comp)
                assert q.forward.input_array.shape == comp.shape
                q.destroy()
                w = newDistArray(pfft, forward_output=spectral, rank=rank, view=True)
                assert isinstance(w, DeviceArray) and not isinstance(w, DistArray) and w.base.rank == rank
        pfft.destroy()
        return True
    assert all(thread_comm.run(2, body))


def test_transfer_accepts_numpy_arrays_and_typecodes():
    """Pencil.transfer(pencilA, pencilB, 'd') called as an unbound method with a typecode, Subcomm
    from None / an int, pencils of lower-dimensional subcomms, host arrays in and out: the calling
    conventions of tests/test_pencil.py:28-55."""
    from tests import thread_comm
    from mpi4py_fft_amd.pencil import Subcomm, Pencil

    def body(comm):
        for shape in ((7, 8), (7, 8, 9)):
            for pdim in [None] + list(range(1, len(shape) - 1)):
                sub = Subcomm(comm, pdim)
                p0 = Pencil(sub, shape)
                pA = p0.pencil(0)
                pB = pA.pencil(1)
                pC = pB.pencil(0 if len(shape) == 2 else -1)
                for code in 'fD':
                    t1, t2 = Pencil.transfer(pA, pB, code), Pencil.transfer(pB, pC, code)
                    X = np.random.default_rng(comm.Get_rank()).random(pA.subshape).astype(code)
                    A, B, C = (np.zeros(p.subshape, dtype=code) for p in (pA, pB, pC))
                    A[...] = X
                    t1.forward(A, B)
                    t2.forward(B, C)
                    B.fill(0)
                    t2.backward(C, B)
                    A.fill(0)
                    t1.backward(B, A)
                    assert np.allclose(A, X)
                    t1.destroy()
                    t2.destroy()
                sub.destroy()
        return True
    assert all(thread_comm.run(3, body))


def test_wisdom_and_timelimit_are_harmless(tmp_path):
    """tests/test_fftw.py:140-160 call these around planning; they must not fail."""
    from mpi4py_fft_amd import fftw
    f = str(tmp_path / 'newwisdom.dat')
    fftw.export_wisdom(f)
    fftw.import_wisdom(f)
    fftw.forget_wisdom()
    fftw.set_timelimit(0.01)
    fftw.cleanup()
