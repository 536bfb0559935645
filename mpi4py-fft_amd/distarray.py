"""Distributed array with device (HBM) backing.

Mirror of mpi4py_fft/distarray.py: ``DistArray`` carries a :class:`Pencil` describing which block
of the global array this rank holds, supports tensor-valued fields (``rank`` leading axes that
are never distributed) and ``redistribute``; ``newDistArray(pfft, ...)`` allocates the input or
output array of a :class:`PFFT`.  The reference subclasses ``np.ndarray``; here the storage is a
:class:`DeviceArray` (see array.py for the ndarray behaviour it keeps).  ``get(gslice)`` gathers a
global slice on rank 0 over the communicator (the reference detours through a parallel HDF5
file); ``write / read`` (HDF5 / NetCDF I/O, distarray.py:365-439) are storage features outside
this path.
"""
from numbers import Number

import numpy as np

from .array import DeviceArray, _item
from .pencil import Pencil, Subcomm
from . import comm as _comm


class DistArray(DeviceArray):
    """Distributed device array (distarray.py:10-101).

    Parameters
    ----------
    global_shape : shape of the non-distributed global array (incl. ``rank`` leading axes)
    subcomm : None, a :class:`Subcomm`, a sequence of communicators, or a sequence of ints
        (grid with 0 = free entries)
    val : fill value;  dtype : 'f', 'd', 'F', 'D';  buffer : a DeviceArray/tensor to adopt
    alignment : the undistributed axis (tensor rank not counted);  rank : tensor rank (0,1,2)
    """
    def __init__(self, global_shape, subcomm=None, val=None, dtype=float, buffer=None,
                 strides=None, alignment=None, rank=0):
        global_shape = tuple(int(s) for s in global_shape)
        field_shape = global_shape[rank:]                 # the distributed part
        self._rank = rank
        if len(field_shape) < 2:                          # 1-D: a plain undistributed array
            self._p0 = None
            self._init_storage(global_shape, dtype, buffer, val)
            return
        subcomm, alignment = self._resolve_grid(subcomm, alignment, len(field_shape))
        self._p0 = Pencil(subcomm, field_shape, axis=alignment)
        self._init_storage(global_shape[:rank] + self._p0.subshape, dtype, buffer, val)

    @staticmethod
    def _resolve_grid(subcomm, alignment, ndim):
        """(Subcomm, aligned axis) from the accepted spellings of `subcomm` (distarray.py:64-94):
        a Subcomm, a sequence of communicators, a grid of ints with 0 = free, or None (everything
        distributed except `alignment`, default the last axis)."""
        if not isinstance(subcomm, Subcomm):
            if subcomm is None:
                grid = [0] * ndim
                alignment = ndim - 1 if alignment is None else alignment
                grid[alignment] = 1
                subcomm = Subcomm(_comm.world(), grid)
            else:
                assert isinstance(subcomm, (tuple, list)) and len(subcomm) == ndim
                if not all(isinstance(c, _comm.Comm) for c in subcomm):
                    subcomm = Subcomm(_comm.world(), subcomm)
        undivided = [i for i, c in enumerate(subcomm) if c.Get_size() == 1]
        if alignment is None:
            alignment = undivided[-1]
        assert isinstance(alignment, (int, np.integer)) and alignment in undivided
        return subcomm, int(alignment)

    def _init_storage(self, shape, dtype, buffer, val):
        tensor = None
        if buffer is not None:
            tensor = buffer.tensor if isinstance(buffer, DeviceArray) else buffer
            tensor = tensor.reshape(tuple(shape))
        DeviceArray.__init__(self, shape, dtype, tensor=tensor,
                             val=val if (buffer is None and isinstance(val, Number)) else None)

    def __getitem__(self, key):
        """A component of a tensor field stays a DistArray of lower rank (``v[0]`` of a rank-1
        field is a rank-0 DistArray that ``PFFT(darray=...)`` accepts); anything that cuts into the
        distributed axes is a plain array view (distarray.py:155-175)."""
        if isinstance(key, DeviceArray):
            key = key.tensor
        sub = self._t[key]
        if sub.ndim == 0:
            return _item(sub)
        lead = None
        if self._p0 is not None and self.ndim > 1:
            if isinstance(key, (int, np.integer, slice)):
                lead = self._rank > 0
            elif isinstance(key, tuple) and all(isinstance(k, (int, np.integer, slice)) for k in key):
                lead = len(key) <= self._rank
        if lead:
            out = DistArray.__new__(DistArray)
            out._p0, out._rank = self._p0, self._rank - (self.ndim - sub.ndim)
        else:
            out = DeviceArray.__new__(DeviceArray)
        out._shape, out._dtype, out._t = tuple(sub.shape), self._dtype, sub
        return out

    # metadata (distarray.py:103-153)
    alignment = property(lambda self: self._p0.axis)
    pencil = property(lambda self: self._p0)
    rank = property(lambda self: self._rank)
    dimensions = property(lambda self: len(self._p0.shape))
    global_shape = property(lambda self: self.shape[:self._rank] + self._p0.shape)
    substart = property(lambda self: (0,) * self._rank + self._p0.substart)
    subcomm = property(lambda self: (_comm.COMM_SELF,) * self._rank + self._p0.subcomm)
    commsizes = property(lambda self: [c.Get_size() for c in self.subcomm])

    @property
    def v(self):
        """The plain array view (``.view(np.ndarray)`` in the reference)."""
        out = DeviceArray.__new__(DeviceArray)
        out._shape, out._dtype, out._t = self._shape, self._dtype, self._t
        out.base = self            # as ndarray views expose the array they were taken from
        return out

    def local_slice(self):
        lead = [slice(0, n) for n in self.shape[:self._rank]]
        block = [slice(s, s + n) for s, n in zip(self._p0.substart, self._p0.subshape)]
        return tuple(lead + block)

    def get_pencil_and_transfer(self, axis):
        p1 = self._p0.pencil(axis)
        return p1, self._p0.transfer(p1, self.dtype)

    def _cached_transfer(self, axis):
        """The Transfer (exchange plan + staging buffers) towards alignment `axis`, kept on this
        array's pencil: the reference builds and frees the MPI datatypes on every redistribute
        (distarray.py:352-361), which costs it little; here a Transfer owns device staging
        buffers, so arrays that are redistributed every time step reuse theirs."""
        cache = self._p0.__dict__.setdefault('_transfers', {})
        key = (self._p0.axis, int(axis), self.dtype.char)
        if key not in cache:
            cache[key] = self.get_pencil_and_transfer(axis)
        return cache[key]

    def redistribute(self, axis=None, out=None):
        """Global redistribution so that `axis` (or `out`'s aligned axis) becomes undivided
        (distarray.py:298-363).  Vector/tensor components are moved one transfer each."""
        if axis is not None:
            # nothing to move: already aligned there, or the target axis is not divided either (both undivided: only
            # the pencil's label changes) -- `self` comes back in both cases, also when `out` was given
            if axis == self.alignment:
                return self
            if isinstance(out, DistArray):
                assert axis == out.alignment, 'axis %d contradicts the alignment of out (%d)' % (axis, out.alignment)
            if self.commsizes[self.rank + axis] == 1:
                self.pencil.axis = axis
                return self
        target = axis
        if out is not None:
            assert isinstance(out, DistArray), 'out must be a DistArray'
            target = out.alignment
        if out is not None:
            assert self.global_shape == out.global_shape, 'global shapes differ: %r and %r' % (self.global_shape, out.global_shape)
            if self.commsizes == out.commsizes:          # same distribution: a plain copy
                out[...] = self
                return out
            # the transfer moves data between the two aligned axes only: every other axis must be cut alike
            for i in (j for j in range(len(self._p0.shape)) if j not in (self.alignment, target)):
                assert self.pencil.subcomm[i] == out.pencil.subcomm[i], 'axis %d is distributed over different groups' % i
                assert self.pencil.subshape[i] == out.pencil.subshape[i], 'axis %d is cut differently' % i
        axis = target
        p1, transfer = self._cached_transfer(axis)
        if out is None:
            out = DistArray(self.global_shape, subcomm=p1.subcomm, dtype=self.dtype,
                            alignment=axis, rank=self.rank)
        src, dst = self.v, out.v
        for comp in np.ndindex(*self.shape[:self._rank]):       # one exchange per field component
            transfer.forward(src[comp] if comp else src, dst[comp] if comp else dst)
        return out

    def get(self, gslice=None):
        """Without arguments: this rank's block as a host array.  With `gslice` (a sequence of
        ``slice(None)`` and ints, one per global axis): that slice of the GLOBAL array, on rank 0
        (None elsewhere), as distarray.py:182-235 -- which routes the gather through a parallel
        HDF5 file; here the ranks' pieces travel over the communicator."""
        if gslice is None:
            return DeviceArray.get(self)
        gslice = tuple(gslice)
        assert len(gslice) == len(self.global_shape)
        mine = self.local_slice()
        kept = [i for i, g in enumerate(gslice) if isinstance(g, slice)]
        index, here = [], True
        for g, blk in zip(gslice, mine):
            if isinstance(g, slice):
                assert g == slice(None), 'only full slices and integer indices'
                index.append(slice(None))
            elif blk.start <= g < blk.stop:
                index.append(int(g) - blk.start)
            else:
                here = False
        piece = np.array(self.v[tuple(index)], copy=True) if here else None    # own memory: ranks move on
        parent = next((c.relay_parent for c in self._p0.subcomm if getattr(c, 'relay_parent', None)), None)
        parts = [(tuple(mine[i] for i in kept), piece)]
        rank0 = True
        if parent is not None:
            parts = parent.allgather_obj(parts[0])
            rank0 = parent.Get_rank() == 0
        if not rank0:
            return None
        out = np.zeros([self.global_shape[i] for i in kept], dtype=self.dtype)
        for where, data in parts:
            if data is not None:
                out[where] = data
        return out

    def write(self, filename, name='darray', step=0, global_slice=None, domain=None, as_scalar=False):
        """Store snapshot `step` of this array (or of ``global_slice`` of it) in an HDF5 (``*.h5``)
        or NetCDF file, or in an open :class:`.io.FileBase` (distarray.py:365-403)."""
        from .io import FileBase, HDF5File, NCFile
        if isinstance(filename, str):
            f = (HDF5File if filename.endswith('.h5') else NCFile)(filename, domain=domain, mode='a')
        else:
            assert isinstance(filename, FileBase)
            f = filename
        field = [self] if global_slice is None else [(self, global_slice)]
        f.write(step, {name: field}, as_scalar=as_scalar)

    def read(self, filename, name='darray', step=0):
        """Fill this array from field `name`, snapshot `step`, of a file written by :meth:`write`
        (whole arrays only; distarray.py:405-439)."""
        from .io import FileBase, HDF5File, NCFile
        if isinstance(filename, str):
            f = (HDF5File if filename.endswith('.h5') else NCFile)(filename, mode='r')
        else:
            assert isinstance(filename, FileBase)
            f = filename
        f.read(self, name, step=step)


def newDistArray(pfft, forward_output=True, val=0, rank=0, view=False):
    """A new DistArray shaped and typed as the input (``forward_output=False``) or output of
    ``pfft.forward`` (distarray.py:442-485)."""
    side = pfft.forward.output_array if forward_output else pfft.forward.input_array
    pencil = pfft.pencil[1 if forward_output else 0]
    field = tuple(pfft.global_shape(bool(forward_output)))
    # a rank-r field carries r leading component axes of length ndim (vectors, tensors), never distributed
    z = DistArray((len(field),) * rank + field, subcomm=pencil.subcomm, val=val, dtype=side.dtype,
                  alignment=pencil.axis, rank=rank)
    return z.v if view else z


def Function(*args, **kwargs):  # deprecated alias kept by the reference (distarray.py:487-493)
    import warnings
    warnings.warn("Function() is deprecated; use newDistArray().", FutureWarning)
    if 'tensor' in kwargs:
        kwargs['rank'] = 1
        del kwargs['tensor']
    return newDistArray(*args, **kwargs)
