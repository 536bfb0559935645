"""The reference's own example scripts (examples/transforms.py, examples/darray.py) with the import
lines switched to this package and MPI.COMM_WORLD replaced by the communicator under test; reductions
over ranks go through allgather_obj.  They exercise the drop-in surface end to end: numpy interop
(np.sum / np.linalg.norm / np.zeros_like / np.allclose on device arrays), `darray=`, rank-1/2
fields, `redistribute(out=)`, `get(gslice)`, `transforms=` with collapse / slab grid / padding."""
import functools

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from tests import cases


def _transforms_example(comm):
    from mpi4py_fft_amd import PFFT, newDistArray
    from mpi4py_fft_amd.fftw import dctn, idctn
    N = np.array([18, 18, 18], dtype=int)
    dct = functools.partial(dctn, type=3)
    idct = functools.partial(idctn, type=3)
    transforms = {(1, 2): (dct, idct)}
    fft = PFFT(comm, N, axes=None, collapse=True, grid=(-1,), transforms=transforms)
    pfft = PFFT(comm, N, axes=((0,), (1, 2)), grid=(-1,), padding=[1.5, 1.0, 1.0], transforms=transforms)
    assert fft.axes == pfft.axes
    u = newDistArray(fft, forward_output=False)
    u[:] = np.random.random(u.shape).astype(u.dtype)
    u_hat = newDistArray(fft, forward_output=True)
    u_hat = fft.forward(u, u_hat)
    uj = np.zeros_like(u)
    uj = fft.backward(u_hat, uj)
    assert np.allclose(uj, u)
    u_padded = newDistArray(pfft, forward_output=False)
    uc = u_hat.copy()
    u_padded = pfft.backward(u_hat, u_padded)
    u_hat = pfft.forward(u_padded, u_hat)
    assert np.allclose(u_hat, uc)
    cfft = PFFT(comm, N, dtype=complex)
    uc = np.random.random(cfft.backward.input_array.shape).astype(complex)
    u2 = cfft.backward(uc)
    u3 = uc.copy()
    u3 = cfft.forward(u2, u3)
    assert np.allclose(uc, u3)
    fft.destroy()
    pfft.destroy()
    cfft.destroy()
    return True


def _darray_example(comm):
    from mpi4py_fft_amd.distarray import DistArray, newDistArray
    from mpi4py_fft_amd.mpifft import PFFT
    from mpi4py_fft_amd import Subcomm
    allreduce = lambda x: sum(comm.allgather_obj(float(x)))

    def darr(N, alignment, **kw):
        nd = len(N) - kw.get('rank', 0)
        grid = [0] * nd
        grid[alignment] = 1
        return DistArray(N, subcomm=Subcomm(comm, grid), alignment=alignment, **kw)
    N = (16, 14, 12)
    z0 = darr(N, 0, dtype=float)
    z0[:] = np.random.randint(0, 10, z0.shape)
    s0 = allreduce(np.sum(z0))
    z1 = z0.redistribute(2)
    s1 = allreduce(np.sum(z1))
    z2 = z1.redistribute(1)
    s2 = allreduce(np.sum(z2))
    assert s0 == s1 == s2
    fft = PFFT(comm, darray=z2, axes=(0, 2, 1))
    z3 = newDistArray(fft, forward_output=True)
    z2c = z2.copy()
    fft.forward(z2, z3)
    fft.backward(z3, z2)
    s0, s1 = np.linalg.norm(z2), np.linalg.norm(z2c)
    assert abs(s0 - s1) < 1e-12, s0 - s1
    v0 = newDistArray(fft, forward_output=False, rank=1)
    v0[:] = np.random.random(v0.shape)
    v0c = v0.copy()
    v1 = newDistArray(fft, forward_output=True, rank=1)
    for i in range(3):
        v1[i] = fft.forward(v0[i], v1[i])
    for i in range(3):
        v0[i] = fft.backward(v1[i], v0[i])
    s0, s1 = np.linalg.norm(v0c), np.linalg.norm(v0)
    assert abs(s0 - s1) < 1e-12
    nfft = PFFT(comm, darray=v0[0], axes=(0, 2, 1))
    for i in range(3):
        v1[i] = nfft.forward(v0[i], v1[i])
    for i in range(3):
        v0[i] = nfft.backward(v1[i], v0[i])
    s0, s1 = np.linalg.norm(v0c), np.linalg.norm(v0)
    assert abs(s0 - s1) < 1e-12
    N = (6, 6, 6)
    z = darr(N, 0, dtype=float)
    z[:] = comm.Get_rank()
    g0 = z.get((0, slice(None), 0))
    z2 = z.redistribute(2)
    z = z2.redistribute(out=z)
    g1 = z.get((0, slice(None), 0))
    assert np.all(g0 == g1)
    s0 = allreduce(np.linalg.norm(z) ** 2)
    s1 = allreduce(np.linalg.norm(z2) ** 2)
    assert abs(s0 - s1) < 1e-12
    N = (3, 3, 6, 6, 6)
    z2 = darr(N, 2, dtype=float, val=1, rank=2)
    z2[:] = comm.Get_rank()
    z1 = z2.redistribute(1)
    z0 = z1.redistribute(0)
    s0 = allreduce(np.linalg.norm(z2) ** 2)
    s1 = allreduce(np.linalg.norm(z0) ** 2)
    assert abs(s0 - s1) < 1e-12
    z1 = z0.redistribute(out=z1)
    z0 = z1.redistribute(out=z0)
    N = (6, 6, 6, 6, 6)
    m0 = darr(N, 2, dtype=float)
    m0[:] = comm.Get_rank()
    m1 = m0.redistribute(4)
    m0 = m1.redistribute(out=m0)
    s0 = allreduce(np.linalg.norm(m0) ** 2)
    s1 = allreduce(np.linalg.norm(m1) ** 2)
    assert abs(s0 - s1) < 1e-12
    return True


@pytest.mark.parametrize('P', [2, 4])     # on one rank collapse merges all three axes (in the reference too)
def test_examples_transforms(P):
    assert all(cases.run_ranks(P, _transforms_example))


@pytest.mark.parametrize('P', [1, 2, 4])
def test_examples_darray(P):
    assert all(cases.run_ranks(P, _darray_example))
