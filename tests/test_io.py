"""Storage boundary: DistArray snapshots in NetCDF / HDF5 files, laid out as the
reference's writers lay them out (io/h5py_file.py, io/nc_file.py; tests/test_io.py there).

NetCDF files are REAL (scipy.io.netcdf_file, NetCDF-3 classic) and are checked by reading them back
with scipy directly.  h5py is not in the image: the HDF5 writer runs against tests/fake_h5py.py,
which checks dataset paths / shapes / regions, not the byte format."""
import sys

import numpy as np
import pytest

N3 = (12, 13, 14)


def _global(shape, seed=7):
    return np.random.default_rng(seed).standard_normal(shape)


def _fill(u, G):
    u[...] = G[u.local_slice()]
    return u


def _nc_run(tmp_path, nranks, grid, rank=0, dtype='d', slab=None):
    """Every rank writes three snapshots (+ slices) of a field cut from one global array; returns
    the per-rank read-back errors."""
    from tests import thread_comm
    from mpi4py_fft_amd import DistArray, Subcomm, NCFile, io
    name = str(tmp_path / ('f%d_%d.nc' % (nranks, rank)))
    shape = (3,) * rank + N3
    G = _global(shape).astype(dtype)
    dom = ((0, np.pi), (0, 2 * np.pi), (0, 3 * np.pi))

    def body(c):
        if slab:
            io.SLAB_BYTES = slab
        sub = Subcomm(c, list(grid))
        u = _fill(DistArray(shape, subcomm=sub, dtype=dtype, rank=rank), G)
        f = NCFile(name, domain=dom, mode='w') if c.Get_rank() == 0 else None
        c.barrier()
        f = f or NCFile(name, domain=dom, mode='a')
        assert f.backend() == 'netcdf4'
        for k in range(3):
            f.write(k, {'u': [u, (u, [slice(None), slice(None), 4]), (u, [5, 5, slice(None)])]})
        f.write(2, {'u': [u]})                                  # same step again: no new record
        u.write(name, 'w', 7, [slice(None), 6, slice(None)])    # by file name, a step that is not an index
        if rank:
            u.write(f, 'u', 1, as_scalar=True)
        u0 = DistArray(shape, subcomm=sub, dtype=dtype, rank=rank)
        f.read(u0, 'u', step=1)
        e1 = np.abs(np.asarray(u0) - np.asarray(u)).max()
        u0[...] = 0
        u0.read(name, 'u', 2)
        return max(e1, np.abs(np.asarray(u0) - np.asarray(u)).max())
    errs = thread_comm.run(nranks, body)
    assert max(errs) == 0.0
    return name, G


@pytest.mark.parametrize('nranks,grid', [(1, (1, 1, 1)), (2, (0, 1, 1)), (4, (0, 0, 1)), (3, (1, 0, 1))])
def test_netcdf_layout_and_values(tmp_path, nranks, grid):
    from scipy.io import netcdf_file
    name, G = _nc_run(tmp_path, nranks, grid)
    f = netcdf_file(name, 'r', mmap=False)
    assert f.dimensions['time'] is None and [f.dimensions[k] for k in 'xyz'] == list(N3)
    assert list(f.variables['time'].data) == [0, 1, 2, 7]
    assert np.allclose(f.variables['x'].data, np.linspace(0, np.pi, N3[0]))
    u = f.variables['u']
    assert u.dimensions == ('time', 'x', 'y', 'z') and u.data.shape[0] == 4
    for k in range(3):
        assert np.array_equal(u.data[k], G)
        assert np.array_equal(f.variables['u_slice_slice_4'].data[k], G[:, :, 4])
        assert np.array_equal(f.variables['u_5_5_slice'].data[k], G[5, 5, :])
    assert f.variables['u_slice_slice_4'].dimensions == ('time', 'x', 'y')
    assert f.variables['u_5_5_slice'].dimensions == ('time', 'z')
    w = f.variables['w_slice_6_slice']
    assert w.dimensions == ('time', 'x', 'z') and np.array_equal(w.data[3], G[:, 6, :])
    f.close()


@pytest.mark.parametrize('rank', [1, 2])
def test_netcdf_tensor_fields(tmp_path, rank):
    from scipy.io import netcdf_file
    name, G = _nc_run(tmp_path, 4, (0, 0, 1), rank=rank, dtype='f', slab=4096)
    f = netcdf_file(name, 'r', mmap=False)
    u = f.variables['u']
    assert u.dimensions == ('time',) + tuple('ij'[:rank]) + ('x', 'y', 'z')
    assert np.array_equal(u.data[0], G)
    comp = (1,) * rank
    scal = f.variables['u' + '1' * rank]
    assert scal.dimensions == ('time', 'x', 'y', 'z') and np.array_equal(scal.data[1], G[comp])
    f.close()


def test_netcdf_refuses_complex_and_bad_names(tmp_path):
    from mpi4py_fft_amd import DistArray, NCFile
    u = DistArray(N3, dtype='D')
    with pytest.raises(TypeError):
        u.write(str(tmp_path / 'c.nc'), 'u', 0)
    v = DistArray(N3, dtype='d')
    with pytest.raises(AssertionError):
        NCFile(str(tmp_path / 'd.nc'), mode='w').write(0, {'x': [v]})


def test_hdf5_needs_h5py(tmp_path):
    from mpi4py_fft_amd import DistArray, HDF5File
    try:
        import h5py  # noqa: F401
        pytest.skip('h5py is present')
    except ImportError:
        pass
    with pytest.raises(ImportError):
        HDF5File(str(tmp_path / 'a.h5'), mode='w')
    with pytest.raises(ImportError):
        DistArray(N3).write(str(tmp_path / 'a.h5'), 'u', 0)


@pytest.mark.parametrize('nranks,grid,rank,dtype', [(1, (1, 1, 1), 0, 'd'), (4, (0, 0, 1), 0, 'D'),
                                                    (4, (0, 1, 0), 1, 'F'), (2, (1, 0, 1), 2, 'd')])
def test_hdf5_layout_on_the_stand_in(tmp_path, monkeypatch, nranks, grid, rank, dtype):
    from tests import fake_h5py, thread_comm
    from mpi4py_fft_amd import DistArray, Subcomm, HDF5File
    monkeypatch.setitem(sys.modules, 'h5py', fake_h5py)
    name = str(tmp_path / 'f.h5')
    shape = (3,) * rank + N3
    G = _global(shape).astype(dtype)
    if dtype in 'FD':
        G = G + 1j * _global(shape, 8).astype(dtype)
    mesh = tuple(np.arange(n, dtype=float) * np.pi / n for n in N3)

    def body(c):
        sub = Subcomm(c, list(grid))
        align = [i for i, g in enumerate(grid) if g == 1][-1]
        u = _fill(DistArray(shape, subcomm=sub, dtype=dtype, rank=rank, alignment=align), G)
        f = HDF5File(name, domain=mesh, mode='w') if c.Get_rank() == 0 else None
        c.barrier()
        f = f or HDF5File(name, domain=mesh, mode='a')
        assert f.backend() == 'hdf5'
        for k in range(2):
            f.write(k, {'u': [u, (u, [slice(None), 4, slice(None)])], 'v': [(u, [slice(None), 5, 5])]})
        u.write(name, 'u', 2)
        if rank:
            f.write(0, {'u': [u]}, as_scalar=True)
        u0 = DistArray(shape, subcomm=sub, dtype=dtype, rank=rank, alignment=align)
        u0.read(name, 'u', 2)
        return np.abs(np.asarray(u0) - np.asarray(u)).max()
    assert max(thread_comm.run(nranks, body)) == 0.0
    f = fake_h5py.File(name, 'r')
    assert sorted(f['u/3D'].keys()) == ['0', '1', '2']
    assert np.array_equal(f['u/3D/1'][...], G)
    assert np.array_equal(f['u/2D/slice_4_slice/1'][...], G[..., :, 4, :])
    assert np.array_equal(f['v/1D/slice_5_5/0'][...], G[..., :, 5, 5])
    assert list(f['u'].attrs['shape']) == list(N3) and int(f['u'].attrs['rank']) == rank
    assert np.array_equal(f['u/mesh/x1'][...], mesh[1])
    if rank:
        comp = (2,) * rank
        assert np.array_equal(f['u' + '2' * rank + '/3D/0'][...], G[comp])


@pytest.mark.gpu
def test_device_arrays_stream_to_netcdf_slab_by_slab(tmp_path, monkeypatch):
    """Blocks in HBM leave through the pinned staging path, several slabs per block."""
    from scipy.io import netcdf_file
    from tests import thread_comm
    from mpi4py_fft_amd import DistArray, Subcomm, io
    monkeypatch.setattr(io, 'SLAB_BYTES', 1 << 20)
    shape = (96, 64, 80)
    G = _global(shape)
    name = str(tmp_path / 'dev.nc')

    def body(c):
        sub = Subcomm(c, [0, 1, 1])
        u = _fill(DistArray(shape, subcomm=sub, dtype='d'), G)
        assert u.tensor.is_cuda
        u.write(name, 'u', 0)
        u.write(name, 'u', 0, [slice(None), 7, slice(None)])
        u0 = DistArray(shape, subcomm=sub, dtype='d')
        u0.read(name, 'u', 0)
        return float(np.abs(np.asarray(u0) - G[u.local_slice()]).max())
    assert max(thread_comm.run(2, body)) == 0.0
    f = netcdf_file(name, 'r', mmap=False)
    assert np.array_equal(f.variables['u'].data[0], G)
    assert np.array_equal(f.variables['u_slice_7_slice'].data[0], G[:, 7, :])
    f.close()


@pytest.mark.parametrize('how', ['comm', 'deferred'])
def test_a_late_rank_cannot_truncate_what_rank_0_has_written(tmp_path, how, monkeypatch):
    """File creation is collective (round 2 let every rank create / truncate on its own: a rank that
    reached the constructor after rank 0 had written wiped rank 0's block).  `comm=`: rank 0 creates,
    barrier.  Without it, in a multi-process world, creation waits for the first write(), where the
    first rank of the array's grid does it at the head of the turn-taking loop."""
    import time
    from tests import thread_comm
    from mpi4py_fft_amd import DistArray, Subcomm, NCFile
    from mpi4py_fft_amd import comm as _comm
    name = str(tmp_path / 'late.nc')
    G = _global((8, 6, 4))
    if how == 'deferred':
        monkeypatch.setattr(_comm, 'world', lambda: type('W', (), {'Get_size': staticmethod(lambda: 2)})())

    def body(c):
        sub = Subcomm(c, [0, 1, 1])
        u = _fill(DistArray((8, 6, 4), subcomm=sub, dtype=float), G)
        if c.Get_rank() == 1:
            time.sleep(0.5)                      # rank 0 is through its constructor -- and, round 2, its write
        f = NCFile(name, mode='w', comm=c) if how == 'comm' else NCFile(name, mode='w')
        f.write(0, {'u': [u]})
        c.barrier()
        back = DistArray((8, 6, 4), subcomm=sub, dtype=float)
        f.read(back, 'u', step=0)
        return np.abs(np.asarray(back) - np.asarray(u)).max()
    assert max(thread_comm.run(2, body)) == 0.0
    from scipy.io import netcdf_file
    nc = netcdf_file(name, 'r', mmap=False)
    assert np.array_equal(nc.variables['u'].data[0], G)
    nc.close()
