"""Drop-in checks modelled on what the reference's example scripts do with the public API
(examples/transforms.py: r2r `transforms=` with collapse / slab grid / per-axis padding and numpy
in/out arrays; examples/darray.py: redistribute chains, `PFFT(darray=...)`, rank-1/2 fields,
`redistribute(out=)`, `get(gslice)`), written against this package's communicators.  What matters
is the numpy interop on device arrays: np.sum / np.linalg.norm / np.zeros_like / np.allclose,
`field[i] = fft.forward(field[i], spec[i])`, and tensor components staying DistArrays."""
import functools

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from tests import cases


def _total(comm, x):
    return sum(comm.allgather_obj(float(x)))


def _field(comm, shape, aligned, **kw):
    """DistArray distributed over every axis but `aligned` (what `DistArray(N, alignment=a)` gives
    on COMM_WORLD in the reference)."""
    from mpi4py_fft_amd import DistArray, Subcomm
    grid = [0] * (len(shape) - kw.get('rank', 0))
    grid[aligned] = 1
    return DistArray(shape, subcomm=Subcomm(comm, grid), alignment=aligned, **kw)


def r2r_slab_roundtrips(comm):
    from mpi4py_fft_amd import PFFT, newDistArray, fftw
    shape = np.array([18, 18, 18], dtype=int)
    pair = (functools.partial(fftw.dctn, type=3), functools.partial(fftw.idctn, type=3))
    plain = PFFT(comm, shape, axes=None, collapse=True, grid=(-1,), transforms={(1, 2): pair})
    padded = PFFT(comm, shape, axes=((0,), (1, 2)), grid=(-1,), padding=[1.5, 1.0, 1.0], transforms={(1, 2): pair})
    assert plain.axes == padded.axes == ((0,), (1, 2))
    x = newDistArray(plain, forward_output=False)
    x[:] = np.random.default_rng(comm.Get_rank()).random(x.shape).astype(x.dtype)
    spec = plain.forward(x, newDistArray(plain, forward_output=True))
    back = plain.backward(spec, np.zeros_like(x))            # numpy output array
    assert isinstance(back, np.ndarray) and np.allclose(back, x)
    keep = spec.copy()
    wide = padded.backward(spec, newDistArray(padded, forward_output=False))
    spec = padded.forward(wide, spec)
    assert np.allclose(spec, keep)                            # pad then truncate = identity
    cplx = PFFT(comm, shape, dtype=complex)
    c0 = np.random.default_rng(1).random(cplx.backward.input_array.shape).astype(complex)
    c1 = cplx.forward(cplx.backward(c0), c0.copy())           # numpy in, numpy out
    assert np.allclose(c0, c1)
    for f in (plain, padded, cplx):
        f.destroy()
    return True


def distarray_tour(comm):
    from mpi4py_fft_amd import PFFT, newDistArray
    # alignment 0 -> 2 -> 1 conserves the sum
    a0 = _field(comm, (16, 14, 12), 0, dtype=float)
    a0[:] = np.random.default_rng(3 + comm.Get_rank()).integers(0, 10, a0.shape)
    a2 = a0.redistribute(2)
    a1 = a2.redistribute(1)
    sums = [_total(comm, np.sum(a)) for a in (a0, a2, a1)]
    assert sums[0] == sums[1] == sums[2]
    # a transform planned from a distributed array, its aligned axis first
    fft = PFFT(comm, darray=a1, axes=(0, 2, 1))
    spec = newDistArray(fft, forward_output=True)
    before = a1.copy()
    fft.forward(a1, spec)
    fft.backward(spec, a1)
    assert abs(np.linalg.norm(a1) - np.linalg.norm(before)) < 1e-12
    # vector fields, component by component, through two differently built plans
    vec = newDistArray(fft, forward_output=False, rank=1)
    vec[:] = np.random.default_rng(9).random(vec.shape)
    vec0 = vec.copy()
    vspec = newDistArray(fft, forward_output=True, rank=1)
    component_plan = PFFT(comm, darray=vec[0], axes=(0, 2, 1))   # vec[0] is itself a DistArray
    for plan in (fft, component_plan):
        for c in range(3):
            vspec[c] = plan.forward(vec[c], vspec[c])
        for c in range(3):
            vec[c] = plan.backward(vspec[c], vec[c])
        assert abs(np.linalg.norm(vec0) - np.linalg.norm(vec)) < 1e-12
    # global slices survive a round trip through another alignment
    z = _field(comm, (6, 6, 6), 0, dtype=float)
    z[:] = comm.Get_rank()
    line0 = z.get((0, slice(None), 0))
    z_other = z.redistribute(2)
    z = z_other.redistribute(out=z)
    line1 = z.get((0, slice(None), 0))
    assert np.all(line0 == line1)
    assert abs(_total(comm, np.linalg.norm(z) ** 2) - _total(comm, np.linalg.norm(z_other) ** 2)) < 1e-12
    # rank-2 tensor field and a 5-D scalar field
    t2 = _field(comm, (3, 3, 6, 6, 6), 2, dtype=float, val=1, rank=2)
    t2[:] = comm.Get_rank()
    t1 = t2.redistribute(1)
    t0 = t1.redistribute(0)
    assert abs(_total(comm, np.linalg.norm(t2) ** 2) - _total(comm, np.linalg.norm(t0) ** 2)) < 1e-12
    t1 = t0.redistribute(out=t1)
    t0 = t1.redistribute(out=t0)
    m0 = _field(comm, (6, 6, 6, 6, 6), 2, dtype=float)
    m0[:] = comm.Get_rank()
    m4 = m0.redistribute(4)
    m0 = m4.redistribute(out=m0)
    assert abs(_total(comm, np.linalg.norm(m0) ** 2) - _total(comm, np.linalg.norm(m4) ** 2)) < 1e-12
    return True


@pytest.mark.parametrize('P', [2, 4])     # on one rank collapse merges all three axes (in the reference too)
def test_r2r_slab_roundtrips(P):
    assert all(cases.run_ranks(P, r2r_slab_roundtrips))


@pytest.mark.parametrize('P', [1, 2, 4])
def test_distarray_tour(P):
    assert all(cases.run_ranks(P, distarray_tour))


def test_multi_gpu_example_script_across_processes():
    """examples/pfft_multi_gpu.py under torch.distributed.run: 4 processes sharing the GPU (gloo),
    the pipelined wire on torch.distributed, real data; its own checks decide."""
    import os
    import re
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GFFT_DIST_BACKEND='gloo')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK'):
        env.pop(k, None)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '4', '--master-addr',
           '127.0.0.1', '--master-port', '29581', os.path.join(root, 'examples', 'pfft_multi_gpu.py'),
           '--size', '128', '--dtype', 'd', '--wire', 'overlap', '--steps', '2']
    res = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600, text=True)
    assert res.returncode == 0, res.stdout[-3000:]
    m = re.search(r'peak \|u_hat\[3,5,7\]\| = ([0-9.]+), round trip max err ([0-9.e+-]+)', res.stdout)
    assert m, res.stdout[-2000:]
    assert abs(float(m.group(1)) - 0.5) < 1e-12 and float(m.group(2)) < 1e-12
    assert "'chunks'" in res.stdout            # the pipelined plan ran
