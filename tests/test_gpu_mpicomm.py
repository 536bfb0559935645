"""`PFFT(MPI.COMM_WORLD, ...)`: an mpi4py communicator at the public boundary, as the reference's
callers pass it (mpifft.py:202-204, pencil.py:64-93), through `comm.MpiComm` -- MPI for topology and
small objects, libgfft's own RCCL communicators for the device buffers, no torch.distributed.

mpi4py is not in the image, so the communicator is the thread-rank MPI emulation that
oracle/make_golden.py runs the REFERENCE on (same `Create_cart` / `Sub` / `bcast` / `gather` calls),
and RCCL -- which refuses two ranks on one device -- is tests/fake_rccl.  TEST INFRASTRUCTURE: the
product never imports either."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from tests import cases
from oracle import pfft_oracle as O
from oracle import make_golden as MG

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope='module', autouse=True)
def fake_rccl():
    src = os.path.join(HERE, 'fake_rccl', 'fake_rccl.cpp')
    so = os.path.join(HERE, 'fake_rccl', 'libfake_rccl.so')
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(['/opt/rocm/bin/hipcc', '-O2', '-std=c++17', '-fPIC', '-shared', '-x', 'hip',
                               '--offload-arch=gfx950', src, '-o', so])
    from mpi4py_fft_amd import _lib
    _lib.check_wire(_lib.lib().gfft_rccl_load(so.encode()))
    yield
    _lib.lib().gfft_rccl_load(None)


def _mpirun(P, fn):
    """mpiexec -n P: fn(MPI.COMM_WORLD) on P thread-ranks of the emulated MPI, each bound to the GPU."""
    import torch

    def body(world):
        torch.cuda.set_device(0)
        return fn(world)
    return MG.mpirun(P, body)


@pytest.mark.parametrize('name', cases.pfft_case_names())
def test_reference_fixtures_through_an_mpi_communicator(name, monkeypatch):
    """The 23 PFFT fixtures the reference produced, with MPI.COMM_WORLD (emulated) as `comm`."""
    monkeypatch.setattr(cases, 'run_ranks', _mpirun)
    cases.check_pfft_golden(name)


@pytest.mark.parametrize('P,shape,dt,kw', [(4, (64, 64, 128), 'D', {}), (8, (64, 64, 64), 'd', {}),
                                           (4, (32, 64, 512), 'F', {}), (2, (64, 32, 64), 'D', dict(grid=(-1,)))])
def test_pipelined_transform_on_an_mpi_communicator(P, shape, dt, kw, monkeypatch):
    """wire='native': the chunked, stream-overlapped pipeline with MPI as nothing but the bootstrap."""
    from mpi4py_fft_amd import PFFT, newDistArray, pipeline, comm
    monkeypatch.setattr(pipeline.Pipeline, 'MIN_CHUNK_BYTES', 0)
    monkeypatch.setattr(pipeline.Pipeline, 'MIN_WIDTH', 4)
    G = O.rng_array(shape, dt, 21)

    def body(world):
        assert not isinstance(world, comm.Comm)            # a foreign (MPI) communicator object
        f = PFFT(world, shape, dtype=dt, wire='native', **kw)
        assert f.pipeline is not None and all(isinstance(c, (comm.MpiComm, comm.SelfComm)) for c in f.subcomm)
        u = newDistArray(f, False)
        u[...] = G[f.local_slice(False)]
        a = np.asarray(f.forward(u)).copy()
        b = np.asarray(f.backward()).copy()
        sl = f.local_slice(False)
        f.destroy()
        return a, b, sl
    res = _mpirun(P, body)
    ref = O.OPFFT(P, shape, dtype=dt, **kw)
    want = ref.forward(ref.scatter(G))
    tol = cases.tol_for(dt)
    for r, (a, b, sl) in enumerate(res):
        assert np.abs(a - want[r]).max() <= tol * np.abs(want[r]).max()
        assert np.abs(b - G[sl]).max() <= 10 * tol * np.abs(G).max()


def test_redistribute_on_an_mpi_communicator():
    """DistArray.redistribute (pencil.Transfer, stand-alone) over MpiComm's gfft_alltoallv."""
    from mpi4py_fft_amd import DistArray, Subcomm
    N = (8, 10, 12)
    G = np.random.default_rng(3).random(N)

    def body(world):
        sub = Subcomm(world, [0, 0, 1])
        z = DistArray(N, subcomm=sub, dtype=float, alignment=2)
        z[...] = G[z.local_slice()]
        z1 = z.redistribute(1)
        return np.array_equal(np.asarray(z1), G[z1.local_slice()]), z1.alignment
    assert all(ok and al == 1 for ok, al in _mpirun(4, body))
