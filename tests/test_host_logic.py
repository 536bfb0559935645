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
    with pytest.raises(NotImplementedError):
        from mpi4py_fft_amd import fftw
        fftw.dctn(None)


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
