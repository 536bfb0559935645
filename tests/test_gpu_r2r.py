"""Real-to-real kinds (DCT / DST I-IV) through gfft_plan_create_r2r, against scipy's unnormalised
transforms -- the pin the reference's own tests use for its FFTW r2r plans
(tests/test_fftw.py:101-118) -- plus the docstring known answers of xfftn.py:378-388,
round trips with get_normalization (tests/test_fftw.py:135-138), mixed kinds per axis, and the
`transforms=` dict of libfft.FFT / PFFT (tests/test_libfft.py:105-108, test_mpifft.py:35-43)."""
import functools
import itertools

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import pfft_oracle as O
from tests import cases


def test_docstring_known_answers():
    from mpi4py_fft_amd import fftw
    A = fftw.aligned(4, dtype='d')
    dct = fftw.dctn(A, flags=(fftw.FFTW_ESTIMATE,))
    A[:] = 1, 2, 3, 4
    B = dct()
    assert np.allclose(B, [20., -6.30864406, 0., -0.44834153], atol=1e-8)
    assert dct.input_array is A and dct.output_array is B
    idct = fftw.idctn(B)
    assert np.allclose(np.asarray(idct()) * fftw.get_normalization(dct.kind, (4,), (0,)), [1, 2, 3, 4], atol=1e-14)
    dst = fftw.dstn(A)
    assert np.allclose(dst(), O.r2r_1d(np.array([1., 2, 3, 4]), 0, fftw.FFTW_RODFT10), atol=1e-13)


@pytest.mark.parametrize('dt', ['d', 'f'])
@pytest.mark.parametrize('typ', [1, 2, 3, 4])
def test_all_kinds_against_scipy(typ, dt):
    from mpi4py_fft_amd import fftw, asdevice
    tol = 1e-12 if dt == 'd' else 5e-5
    for shape, axes in (((16,), (0,)), ((5, 12), (1,)), ((9, 8), (0,)), ((4, 7, 6), (1,)), ((3, 30, 4), (1,)),
                        ((6, 5, 4), (0, 1, 2)), ((4, 8, 16), (2, 1)), ((2, 257), (1,)), ((2, 1000), (1,)),
                        ((1024, 3), (0,)), ((3, 67), (1,)),
                        # logical lengths 64 ... 4096 = powers of two: one register-kernel pass with
                        # the pre / post steps as load / store adapters (rows and strided mappings;
                        # 33 / 31 points make N = 64 for the type-1 kinds)
                        ((5, 64), (1,)), ((64, 5), (0,)), ((3, 128, 70), (1,)), ((2, 33), (1,)), ((2, 31), (1,)),
                        ((33, 70), (0,)), ((31, 3), (0,)), ((3, 2048), (1,)), ((512, 6), (0,)), ((32, 64, 128), (0, 1, 2)),
                        # ... and 3^b 2^k / 5^c 2^k logical lengths (n = 48, 768, 500, 10; 49 / 47 for type 1)
                        ((3, 48), (1,)), ((768, 4), (0,)), ((2, 500), (1,)), ((70, 10), (1,)), ((49, 5), (0,)),
                        ((2, 47), (1,)), ((96, 80, 72), (0, 1, 2))):
        A = O.rng_array(shape, dt, 3)
        for planner, iplanner, table in ((fftw.dctn, fftw.idctn, fftw.dct_type), (fftw.dstn, fftw.idstn, fftw.dst_type)):
            a = asdevice(A)
            plan = planner(a, axes=axes, type=typ)
            want = A.astype('d')
            for ax in reversed(axes):          # last axis first, as any separable plan
                want = O.r2r_1d(want, ax, table[typ])
            got = np.asarray(plan())
            assert got.dtype == A.dtype and got.shape == A.shape
            assert np.abs(got - want).max() <= tol * np.abs(want).max(), (planner.__name__, typ, shape, axes)
            # inverse: idct(dct(x)) * M == x   (tests/test_fftw.py:135-138)
            inv = iplanner(plan.output_array, axes=axes, type=typ)
            M = fftw.get_normalization(plan.kind, shape, axes)
            back = np.asarray(inv()) * M
            assert np.abs(back - A).max() <= tol * 10 * max(1.0, np.abs(A).max()), (planner.__name__, typ, shape, axes)
            assert inv.kind == tuple(fftw.inverse[k] for k in plan.kind)
            plan.destroy()
            inv.destroy()


def test_mixed_kinds_per_axis_and_in_place():
    from mpi4py_fft_amd import fftw, asdevice
    shape, axes = (6, 8, 10), (0, 1, 2)
    rng = np.random.default_rng(1)
    A = rng.standard_normal(shape)
    for kinds in itertools.islice(itertools.product(range(3, 11), repeat=3), 0, None, 37):
        a = asdevice(A)
        plan = fftw.get_planned_FFT(a, a, axes, np.array(kinds), 1, (fftw.FFTW_ESTIMATE,), 1.0)   # in place, numpy kinds
        want = A
        for ax, k in zip(reversed(axes), reversed(kinds)):
            want = O.r2r_1d(want, ax, k)
        got = np.asarray(plan())
        assert np.abs(got - want).max() <= 1e-12 * np.abs(want).max(), kinds
        plan.destroy()


def test_r2r_refusals():
    from mpi4py_fft_amd import fftw, asdevice
    a = asdevice(np.ones(1))
    with pytest.raises(RuntimeError):              # REDFT00 needs two points (FFTW returns a NULL plan)
        fftw.get_planned_FFT(a, asdevice(np.ones(1)), (0,), [fftw.FFTW_REDFT00])
    with pytest.raises(NotImplementedError):
        fftw.get_planned_FFT(asdevice(np.ones(8)), asdevice(np.ones(8)), (0,), [fftw.FFTW_R2HC])


@pytest.mark.parametrize('dt', ['d', 'f'])
def test_libfft_transforms_dict(dt):
    """libfft.FFT(shape, axes, transforms={axes: (dctn, idctn)}) (tests/test_libfft.py:100-125)."""
    from mpi4py_fft_amd import FFT, fftw, asdevice
    dctn = functools.partial(fftw.dctn, type=3)
    idctn = functools.partial(fftw.idctn, type=3)
    tol = 1e-12 if dt == 'd' else 1e-4
    for shape, axes in (((12, 10), (1,)), ((12, 10), (0, 1)), ((6, 8, 10), (1, 2))):
        fft = FFT(shape, axes, dtype=dt, transforms={tuple(axes): (dctn, idctn)})
        ref = O.OFFT(shape, axes, dt, r2r=fftw.FFTW_REDFT01)
        A = O.rng_array(shape, dt, 2)
        B = np.asarray(fft.forward(asdevice(A)))
        want = ref.forward(A)
        assert B.dtype == A.dtype and np.abs(B - want).max() <= tol * max(1e-30, np.abs(want).max())
        A2 = np.asarray(fft.backward(asdevice(want)))
        assert np.abs(A2 - A).max() <= tol * 10
        fft.destroy()


@pytest.mark.parametrize('P', [1, 2, 4])
def test_pfft_with_r2r_transforms(P):
    """PFFT(..., transforms={(1, 2): (dctn, idctn)}) with a Fourier stage on the remaining axis
    (tests/test_mpifft.py:35-57,97-125), forward values and round trip vs the oracle."""
    from mpi4py_fft_amd import PFFT, newDistArray, fftw
    dct = functools.partial(fftw.dctn, type=3)
    idct = functools.partial(fftw.idctn, type=3)
    dst = functools.partial(fftw.dstn, type=2)
    idst = functools.partial(fftw.idstn, type=2)
    for shape, axes, tr, spec in (
            ((12, 14, 16), ((0,), (1, 2)), {(1, 2): (dct, idct)}, {(1, 2): fftw.FFTW_REDFT01}),
            ((12, 14, 16), ((0,), (1,), (2,)), {(2,): (dst, idst), (1,): (dct, idct)}, {(2,): fftw.FFTW_RODFT10, (1,): fftw.FFTW_REDFT01}),
            ((8, 9, 10, 12), ((0,), (1, 2), (3,)), {(1, 2): (dct, idct), (3,): (dst, idst)}, {(1, 2): fftw.FFTW_REDFT01, (3,): fftw.FFTW_RODFT10}),
            ((12, 16), ((0,), (1,)), {(0,): (dct, idct), (1,): (dct, idct)}, {(0,): fftw.FFTW_REDFT01, (1,): fftw.FFTW_REDFT01}),
            # the configuration of tests/test_mpifft.py:35-43 (5-D, slab, DCT-III and DST-III pairs)
            ((5, 6, 7, 8, 9), ((0,), (1, 2), (3, 4)),
             {(1, 2): (dct, idct), (3, 4): (functools.partial(fftw.dstn, type=3), functools.partial(fftw.idstn, type=3))},
             {(1, 2): fftw.FFTW_REDFT01, (3, 4): fftw.FFTW_RODFT01})):
        grid = dict(grid=(-1,)) if len(shape) == 5 else {}
        ref = O.OPFFT(P, shape, axes=axes, dtype='d', r2r=spec, **grid)
        G = O.rng_array(shape, 'd', 11)
        want = ref.forward(ref.scatter(G))

        def body(comm):
            fft = PFFT(comm, shape, axes=axes, dtype='d', transforms=tr, **grid)
            u = newDistArray(fft, False)
            u[...] = G[fft.local_slice(False)]
            uh = np.asarray(fft.forward(u)).copy()
            back = np.asarray(fft.backward()).copy()
            sl = fft.local_slice(False)
            fft.destroy()
            return uh, back, sl
        for r, (uh, back, sl) in enumerate(cases.run_ranks(P, body)):
            assert uh.shape == want[r].shape and uh.dtype == want[r].dtype, (shape, axes)
            assert np.abs(uh - want[r]).max() <= 1e-12 * max(1e-30, np.abs(want[r]).max()), (shape, axes)
            assert np.abs(back - G[sl]).max() <= 1e-12, (shape, axes)


def test_hermitian_planners():
    """hfftn / ihfftn (xfftn.py:616-761): docstring known answers and the round trip of
    tests/test_fftw.py:77-83 (explicit input array, implicit=False, normalize=True)."""
    from mpi4py_fft_amd import fftw
    A = fftw.aligned(4, dtype='D')
    h = fftw.hfftn(A, flags=(fftw.FFTW_ESTIMATE,))
    A[:] = 1, 2, 3, 4
    assert np.allclose(h(), [15., -4., 0., -1., 0., -4.], atol=1e-13)
    h7 = fftw.hfftn(A, s=(7,), flags=(fftw.FFTW_ESTIMATE,))
    A[:] = 1, 2, 3, 4
    assert np.allclose(h7(), [19., -5.04891734, -0.30797853, -0.64310413, -0.64310413, -0.30797853, -5.04891734], atol=1e-7)
    assert h7.input_array is A
    R = fftw.aligned(4, dtype='d')
    ih = fftw.ihfftn(R, flags=(fftw.FFTW_ESTIMATE,))
    R[:] = 1, 2, 3, 4
    assert np.allclose(ih(), [10, -2 + 2j, -2], atol=1e-13)
    for shape, axes in (((7, 8), (1,)), ((8, 10, 7), (2, 1))):
        x = np.random.default_rng(2).random(shape)
        r = fftw.aligned(shape, dtype='d')
        fwd = fftw.rfftn(r, None, axes)
        r[:] = x
        B = np.asarray(fwd()).copy()
        sa = np.take(shape, axes) if shape[axes[-1]] % 2 == 1 else None
        hp = fftw.hfftn(fwd.output_array, sa, axes, output_array=r)
        hp.input_array[...] = B
        AC = hp().copy()
        ip = fftw.ihfftn(r, None, axes, output_array=fwd.output_array)
        A2 = ip(AC, implicit=False, normalize=True)
        assert np.allclose(A2, B, atol=1e-12)
