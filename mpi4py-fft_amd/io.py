"""Storage boundary of the path: snapshots of device-resident DistArrays in HDF5 / NetCDF files.

Counterpart of ``mpi4py_fft/io/{file_base,h5py_file,nc_file}.py`` and of ``DistArray.write / read``
(distarray.py:365-439) with the same file layouts, so that files written on either side are read
by the other:

* HDF5 (h5py_file.py:147-152): dataset ``<name>/<d>D/<step>`` of the GLOBAL shape, sliced snapshots
  under ``<name>/<d>D/<slice_4_slice>/<step>``, group attributes ``shape`` / ``rank`` and the
  ``domain`` / ``mesh`` sub-groups (``x0``, ``x1`` ...).
* NetCDF (nc_file.py:52-206): unlimited dimension ``time``, one dimension + coordinate variable per
  axis (``x y z r s t``, ``i j k`` for tensor components), one record variable per field
  (``u(time, x, y, z)``, ``u_slice_4_slice(time, x, z)``).

What is different, and why.  The reference's ranks write their blocks through MPI-IO (h5py
``driver='mpio'``, netCDF4 ``parallel=True``).  Here a rank is a process that owns one GPU and there is
no MPI underneath, so the ranks of the array's grid take turns: rank r opens the file, writes its
block into its region of the global dataset, closes, and a barrier hands the file to rank r + 1.
The blocks come out of HBM slab by slab along the first local axis (``SLAB_BYTES`` each, through the
pinned staging of array.py), so a 16 GiB block never needs a 16 GiB host copy.

Containers.  ``HDF5File`` needs ``h5py`` (any build: no MPI driver required); it is absent from the
build image, so the class raises ImportError there -- loudly, nothing is silently skipped.
``NCFile`` uses ``netCDF4`` when importable and otherwise writes NetCDF-3 classic (64-bit offsets)
through ``scipy.io.netcdf_file``, which every netCDF reader opens; that container has no complex
type, exactly like the reference's (``createVariable`` refuses complex dtypes).
"""
import os

import numpy as np

from .array import DeviceArray

__all__ = ('FileBase', 'HDF5File', 'NCFile')

SLAB_BYTES = 256 << 20


def _grid_comm(u):
    """The communicator spanning every rank that holds a block of `u` (None: one rank)."""
    p0 = getattr(u, '_p0', None)
    if p0 is None:
        return None
    for c in p0.subcomm:
        parent = getattr(c, 'relay_parent', None)
        if parent is not None and parent.Get_size() > 1:
            return parent
    big = [c for c in p0.subcomm if c.Get_size() > 1]
    assert len(big) <= 1, 'a grid with several divided axes has a parent communicator'
    return big[0] if big else None


def _host_slabs(u, index):
    """Yield (offset along the first kept axis, host block) for ``u[index]``, SLAB_BYTES at a time.
    `index` holds slices and ints (local coordinates); device blocks are staged through pinned
    memory by ``DeviceArray.get``."""
    view = u[tuple(index)] if any(not (isinstance(i, slice) and i == slice(None)) for i in index) else u
    if not isinstance(view, DeviceArray):
        yield 0, np.asarray(view)
        return
    if view.ndim == 0 or view.size == 0:
        yield 0, np.asarray(view.get())
        return
    row = max(1, view.size // view.shape[0]) * view.dtype.itemsize
    step = max(1, SLAB_BYTES // row)
    for a in range(0, view.shape[0], step):
        yield a, np.ascontiguousarray(view[a:a + step].get())


class FileBase(object):
    """Shared logic of the writers (file_base.py:9-140): which dataset a field goes to, which part
    of it this rank owns, and whose turn it is.

    Parameters
    ----------
    filename : str
    domain : sequence, optional
        per axis either ``(origin, length)`` or an array of coordinates
    """
    def __init__(self, filename=None, domain=None):
        self.f = None
        self.filename = filename
        self.domain = domain
        self._pending_create = None

    def _init_file(self, mode, comm, create):
        """Creation / truncation is COLLECTIVE, as the reference's constructor is (its ranks open the
        file together through MPI-IO, h5py_file.py:31-40 / nc_file.py:36-44): exactly one rank creates
        or truncates, and nobody writes before it has.  Every rank doing it on its own -- round 2 --
        let a rank that arrived late truncate what rank 0 had already written.
          * `comm` given: rank 0 of it creates now, then a barrier;
          * one process in the world: create now;
          * otherwise the file is created at the first write(), by rank 0 of the written array's grid
            at the head of the turn-taking loop (the constructor does not know the grid)."""
        from . import comm as _comm
        if mode == 'r':
            return
        want = (lambda: mode == 'w' or not os.path.exists(self.filename))
        if comm is not None and comm.Get_size() > 1:
            if comm.Get_rank() == 0 and want():
                create()
            comm.barrier()
        elif comm is not None or _comm.world().Get_size() == 1:
            if want():
                create()
        else:
            self._pending_create = (mode, create)

    def _create_if_pending(self, first):
        """Head of a turn-taking loop: the first rank of the grid creates the file if that is still due."""
        pc, self._pending_create = self._pending_create, None
        if pc is not None and first and (pc[0] == 'w' or not os.path.exists(self.filename)):
            pc[1]()

    # ---- container hooks ---------------------------------------------------------------------------
    @staticmethod
    def backend():
        raise NotImplementedError

    def open(self, mode='r+'):
        raise NotImplementedError

    def close(self):
        if self.f is not None:
            self.f.close()
            self.f = None

    def _check_domain(self, group, field):
        raise NotImplementedError

    def _dataset(self, name, field, step, kept, slname):
        """Create-or-open the dataset of (name, step) for the global axes `kept`; returns an object
        with region assignment and the index prefix (the NetCDF record index)."""
        raise NotImplementedError

    # ---- turn taking ---------------------------------------------------------------------------------
    def _in_turn(self, comm, body, mode='r+'):
        """Run body() on every rank of `comm`, one rank at a time in rank order, each inside its own
        open / close of the file (the reference's collective open under MPI-IO)."""
        if comm is None or comm.Get_size() == 1:
            if self._pending_create is not None:
                # a world of several processes, no communicator given to the constructor, and an array that is not
                # distributed: every process would create (truncate) the file on its own -- the race _init_file
                # exists to prevent.  Nobody can know here which process should.
                raise RuntimeError('%s: the file still has to be created, the world has several processes and the array '
                                   'written is not distributed; pass comm= (e.g. comm.COMM_SELF for one file per process) to '
                                   'the constructor so that exactly one process creates it' % self.filename)
            self.open(mode)
            try:
                return body()
            finally:
                self.close()
        out = None
        for r in range(comm.Get_size()):
            if r == comm.Get_rank():
                self._create_if_pending(r == 0)
                self.open(mode)
                try:
                    out = body()
                finally:
                    self.close()
            comm.barrier()
        return out

    # ---- writing -------------------------------------------------------------------------------------
    def write(self, step, fields, **kw):
        """Store snapshot `step` of `fields` = ``{name: [array | (array, global_slice), ...]}``
        (file_base.py:36-79).  ``as_scalar=True`` stores the components of tensor fields as separate
        scalar fields ``name0, name1, name01 ...``."""
        as_scalar = kw.get('as_scalar', False)
        jobs = []
        for group, items in fields.items():
            assert isinstance(items, (tuple, list))
            assert isinstance(group, str)
            for item in items:
                u, sl = (item[0], item[1]) if isinstance(item, (tuple, list)) else (item, None)
                if not as_scalar or u.rank == 0:
                    jobs.append((group, u, sl))
                else:
                    for comp in np.ndindex(*u.shape[:u.rank]):
                        jobs.append((group + ''.join(str(k) for k in comp), u[comp], sl))
        if not jobs:
            return
        comm = _grid_comm(jobs[0][1])

        def body():
            it = self._record_index(step)
            for group, u, sl in jobs:
                self._check_domain(group, u)
                self._write_field(group, u, sl, it)
        self._in_turn(comm, body)

    def _record_index(self, step):
        return step

    def _write_field(self, name, u, gslice, step):
        rank = u.rank
        mine = u.local_slice()                               # global extent of this rank's block
        if gslice is None:
            gslice = (slice(None),) * u.dimensions
            slname = None
        else:
            gslice = tuple(gslice)
            assert len(gslice) == u.dimensions
            slname = self._get_slice_name(gslice)
        gslice = (slice(None),) * rank + gslice
        index, region, kept, here = [], [], [], True
        for ax, (g, blk) in enumerate(zip(gslice, mine)):
            if isinstance(g, slice):
                assert g == slice(None), 'global slices are made of slice(None) and integers'
                index.append(slice(None))
                region.append(blk)
                kept.append(ax)
            elif blk.start <= g < blk.stop:
                index.append(int(g) - blk.start)
            else:
                here = False
        dset, prefix = self._dataset(name, u, step, kept, slname)
        if not here:
            return
        if not region:                                       # a single point
            for _, host in _host_slabs(u, index):
                dset[prefix if prefix else ()] = host
            return
        for a, host in _host_slabs(u, index):
            first = slice(region[0].start + a, region[0].start + a + host.shape[0])
            dset[prefix + (first,) + tuple(region[1:])] = host

    # ---- reading -------------------------------------------------------------------------------------
    def read(self, u, name, **kw):
        """Fill `u` (whole arrays only, as in the reference) from field `name`, snapshot ``step``."""
        step = kw.get('step', 0)

        def body():
            dset, prefix = self._existing(name, u, step)
            mine = tuple(u.local_slice())
            row = max(1, u.size // max(1, u.shape[0])) * u.dtype.itemsize
            n = max(1, SLAB_BYTES // row)
            for a in range(0, u.shape[0], n):
                first = slice(mine[0].start + a, min(mine[0].start + a + n, mine[0].stop))
                block = np.asarray(dset[prefix + (first,) + mine[1:]], dtype=u.dtype)
                u[a:a + block.shape[0]] = block
        # readers do not modify the file: no turn taking needed
        self.open('r')
        try:
            body()
        finally:
            self.close()

    def _existing(self, name, u, step):
        raise NotImplementedError

    @staticmethod
    def _get_slice_name(slices):
        return '_'.join('slice' if isinstance(s, slice) else str(s) for s in slices)


class HDF5File(FileBase):
    """HDF5 snapshots in the reference's layout (h5py_file.py:9-152).  ``mode``: r / w / a."""
    def __init__(self, h5name, domain=None, mode='a', comm=None, **kw):
        FileBase.__init__(self, h5name, domain=domain)
        self._h5py = self._import()
        self._kw = kw
        self._init_file(mode, comm, lambda: self._h5py.File(h5name, 'w', **kw).close())

    @staticmethod
    def _import():
        try:
            import h5py
        except ImportError as e:
            raise ImportError('HDF5File needs h5py (not part of the ROCm build image); NCFile writes '
                              'NetCDF files without extra packages') from e
        return h5py

    @staticmethod
    def backend():
        return 'hdf5'

    def open(self, mode='r+'):
        self.f = self._h5py.File(self.filename, mode)

    def _check_domain(self, group, field):
        if self.domain is None:
            self.domain = ((0, 2 * np.pi),) * field.dimensions
        assert len(self.domain) == field.dimensions
        g = self.f.require_group(group)
        if 'shape' not in g.attrs:
            g.attrs.create('shape', field.pencil.shape)
        if 'rank' not in g.attrs:
            g.attrs.create('rank', field.rank)
        assert field.rank == g.attrs['rank']
        assert np.all(np.asarray(field.pencil.shape) == np.asarray(g.attrs['shape']))
        for i, d in enumerate(self.domain):
            if isinstance(d, np.ndarray):
                sub, d0 = g.require_group('mesh'), np.squeeze(d)
            else:
                sub, d0 = g.require_group('domain'), np.array([d[0], d[1]])
            sub.require_dataset('x%d' % i, shape=d0.shape, dtype=d0.dtype, data=d0)

    def _path(self, name, ndims, slname):
        parts = [name, '%dD' % ndims] + ([slname] if slname else [])
        return '/'.join(parts)

    def _dataset(self, name, field, step, kept, slname):
        ndims = len([k for k in kept if k >= field.rank])
        group = self.f.require_group(self._path(name, ndims, slname))
        shape = tuple(field.global_shape[k] for k in kept)
        return group.require_dataset(str(step), shape=shape, dtype=field.dtype), ()

    def _existing(self, name, u, step):
        return self.f['/'.join((name, '%dD' % u.dimensions, str(step)))], ()


class _NC3Variable:
    """Region assignment on a scipy.io.netcdf_file variable.  Its own __setitem__ grows the record
    axis with np.resize (which repeats old records); here new records start as zeros, as the
    reference's collective ``h[step] = 0`` leaves them (nc_file.py:169)."""
    def __init__(self, var):
        self.var = var

    def records(self):
        return len(self.var.data) if self.var.isrec else None

    def grow(self, need):
        v = self.var
        if v.isrec and need > len(v.data):
            grown = np.zeros((need,) + v.data.shape[1:], dtype=v.data.dtype)
            grown[:len(v.data)] = v.data
            v.__dict__['data'] = grown          # (attribute assignment would be stored as a file attribute)

    def __setitem__(self, key, value):
        if isinstance(key, tuple) and key and isinstance(key[0], (int, np.integer)):
            self.grow(int(key[0]) + 1)
        self.var.data[key] = value

    def __getitem__(self, key):
        return self.var.data[key]


class _NC4Variable:
    """The same three operations on a netCDF4 variable."""
    def __init__(self, var):
        self.var = var

    def records(self):
        return self.var.shape[0]

    def grow(self, need):
        for it in range(self.var.shape[0], need):
            self.var[it] = 0

    def __setitem__(self, key, value):
        self.var[key] = value

    def __getitem__(self, key):
        return self.var[key]


class NCFile(FileBase):
    """NetCDF snapshots in the reference's layout (nc_file.py:12-206)."""
    _AXES = 'xyzrst'
    _COMP = 'ijk'

    def __init__(self, ncname, domain=None, mode='a', clobber=True, comm=None, **kw):
        FileBase.__init__(self, ncname, domain=domain)
        try:
            import netCDF4
            self._nc4 = netCDF4
        except ImportError:
            self._nc4 = None
        self.dims = None
        self._init_file(mode, comm, self._create)

    # NOTE on the scipy container: mode 'a' loads the file and rewrites it on close; fine for
    # snapshots of a few GB, use netCDF4 / HDF5File beyond that.
    def _create(self):
        if self._nc4 is not None:
            f = self._nc4.Dataset(self.filename, mode='w')
            f.createDimension('time', None)
            f.createVariable('time', float, ('time',))
        else:
            from scipy.io import netcdf_file
            f = netcdf_file(self.filename, 'w', version=2)
            f.createDimension('time', None)
            f.createVariable('time', 'd', ('time',))
        f.close()

    @classmethod
    def backend(cls):
        return 'netcdf4'

    def open(self, mode='r+'):
        if self._nc4 is not None:
            self.f = self._nc4.Dataset(self.filename, mode=mode)
        else:
            from scipy.io import netcdf_file
            self.f = netcdf_file(self.filename, 'r' if mode == 'r' else 'a', mmap=False)

    def _var(self, name):
        v = self.f.variables[name]
        return _NC4Variable(v) if self._nc4 is not None else _NC3Variable(v)

    def _record_index(self, step):
        """Index along `time` that holds snapshot `step` (nc_file.py:133-140): an earlier snapshot of
        the same step is overwritten, otherwise a record is appended."""
        t = self.f.variables['time']
        have = np.asarray(t[:] if self._nc4 is not None else t.data)
        hit = np.nonzero(have == step)[0]
        if len(hit):
            return int(hit[0])
        self._var('time')[(len(have),)] = step
        return len(have)

    def _check_domain(self, group, field):
        N = field.global_shape[field.rank:]
        if self.domain is None:
            self.domain = [np.linspace(0, 2 * np.pi, n) for n in N]
        assert len(self.domain) == field.dimensions
        if len(self.domain[0]) == 2 and N[0] != 2:
            self.domain = [np.linspace(d[0], d[1], n) for d, n in zip(self.domain, N)]
        self.dims = ['time']
        for i in range(field.rank):
            ind = self._COMP[i]
            self.dims.append(ind)
            if ind not in self.f.variables:
                self.f.createDimension(ind, field.dimensions)
                self.f.createVariable(ind, 'd', (ind,))[:] = np.arange(field.dimensions)
        for i in range(field.dimensions):
            ax = self._AXES[i]
            self.dims.append(ax)
            if ax not in self.f.variables:
                self.f.createDimension(ax, N[i])
                self.f.createVariable(ax, 'd', (ax,))[:] = self.domain[i]

    def _dataset(self, name, field, step, kept, slname):
        assert name not in self.dims, 'field names x, y, z ... are taken by the axes'
        if np.dtype(field.dtype).kind == 'c':
            raise TypeError('NetCDF has no complex type (store .real / .imag, or use HDF5File)')
        fname = name if slname is None else '_'.join((name, slname))
        if fname not in self.f.variables:
            dims = ['time'] + [self.dims[k + 1] for k in kept]
            self.f.createVariable(fname, np.dtype(field.dtype).char, tuple(dims))
        var = self._var(fname)
        var.grow(int(step) + 1)                  # every rank, also those that hold no part of a slice
        return var, (int(step),)

    def _existing(self, name, u, step):
        # `step` indexes the RECORD directly, as the reference's read does (nc_file.py: `self.f[name][(step,) + s]`)
        # -- write() is what maps a step value to a record through the `time` variable (nc_file.py:150-156),
        # so the two agree for steps written in order 0, 1, 2, ... and differ, there as here, otherwise
        return self._var(name), (int(step),)


def generate_xdmf(h5filename, periodic=True, order='visit'):
    """Name kept for scripts written against the reference (mpi4py_fft/__init__.py:26 exports it;
    io/generate_xdmf.py writes XDMF visualisation metadata next to an HDF5 snapshot file).  Not rebuilt
    here: it is pure host-side XML generation outside the PFFT hot path (DESIGN.md
    section 7) -- the reference's own generator reads the files HDF5File writes (same dataset layout,
    io/h5py_file.py:147-152) and can be used unchanged."""
    raise NotImplementedError('generate_xdmf is not part of this package: run mpi4py_fft.io.generate_xdmf '
                              'on the HDF5 file (the dataset layout is the reference\'s)')

