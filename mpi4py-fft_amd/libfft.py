"""Serial (per-rank) multi-axis transform object on device memory.

Mirror of ``mpi4py_fft.libfft.FFT`` with its fftw backend (libfft.py:314-434): owns the input
array U and output array V of one axis group, ``forward``/``backward`` callables with the
reference's conventions -- forward scaled by ``M = 1/prod(N_axes)`` unless ``normalize=False``,
backward unscaled unless ``normalize=True``; real input -> r2c along ``axes[-1]``; optional
3/2-rule padding (truncate after forward, zero-pad before backward).

MI355X differences, none visible in results: the ``*= M`` pass of libfft.py:412-413 is fused into
the last FFT kernel's store (or into the truncation kernel when padding), and the
truncation/padding strided numpy copies (libfft.py:263-311) are single HIP kernels.
"""
import numpy as np

from . import fftw
from . import _lib
from .array import DeviceArray


class _Xfftn_wrap:
    # Callable with ``input_array`` / ``output_array`` attributes (libfft.py:187-219).
    __slots__ = ('_xfftn', '_input_array', '_output_array')

    def __init__(self, xfftn_obj, input_array, output_array):
        self._xfftn = xfftn_obj
        self._input_array = input_array
        self._output_array = output_array

    input_array = property(lambda self: self._input_array)
    output_array = property(lambda self: self._output_array)
    xfftn = property(lambda self: self._xfftn)

    def __call__(self, input_array=None, output_array=None, **options):
        if input_array is not None:
            self._input_array[...] = input_array
        self._xfftn(**options)
        if output_array is not None:
            output_array[...] = self._output_array
            return output_array
        return self._output_array


class _ShapeOnly:
    """Stand-in for an array that is only planned against (shape / dtype / strides), never touched."""
    def __init__(self, shape, dtype):
        self.shape = tuple(int(n) for n in shape)
        self.dtype = np.dtype(dtype)
        st, acc = [], self.dtype.itemsize
        for n in reversed(self.shape):
            st.insert(0, acc)
            acc *= n
        self.strides = tuple(st)


def _as_list(x):
    return [int(v) for v in x] if np.ndim(x) else [int(x)]


def _normalise_axes(shape, axes):
    """(shape, axes) as lists: every extent positive, axes wrapped into 0 .. ndim - 1, each named at most once, at
    least one and at most ndim of them; axes=None means all, in order.  The argument contract of the reference's
    serial transforms (mpi4py_fft/libfft.py:238-255), violations raise AssertionError as they do there."""
    shape = _as_list(shape)
    ndim = len(shape)
    assert ndim > 0, 'shape must name at least one axis'
    assert all(n > 0 for n in shape), 'every extent must be positive, got %r' % (shape,)
    if axes is None:
        return shape, list(range(ndim))
    axes = [a + ndim if a < 0 else a for a in _as_list(axes)]
    assert 0 < len(axes) <= ndim, '%d axes named for a %d-dimensional array' % (len(axes), ndim)
    assert all(0 <= a < ndim for a in axes), 'axis out of range for a %d-dimensional array: %r' % (ndim, axes)
    assert len(set(axes)) == len(axes), 'an axis is named twice: %r' % (axes,)
    return shape, axes


class FFTBase:
    """Argument normalisation shared by serial transforms (libfft.py:221-261)."""
    def __init__(self, shape, axes=None, dtype=float, padding=False):
        shape, axes = _normalise_axes(shape, axes)
        dtype = np.dtype(dtype)
        assert dtype.char in 'fdgFDG', 'dtype must be a real or complex floating type, got %s' % dtype
        if dtype.char in 'gG':
            raise NotImplementedError('long double transforms have no MI355X type (fp32 and fp64 only)')
        self.shape = shape
        self.axes = axes
        self.dtype = dtype
        self.padding = padding
        self.real_transform = np.issubdtype(dtype, np.floating)
        self.padding_factor = 1


class FFT(FFTBase):
    """Serial transform over ``axes`` of device arrays of ``shape``.

    Parameters as the reference (libfft.py:376-377).  ``backend`` is accepted for signature
    compatibility: 'fftw' (the reference's default name) and 'gfft' both select the one engine
    this package has; any other name raises.  ``transforms`` may map axes to
    ``(fftw.fftn, fftw.ifftn)`` / ``(fftw.rfftn, fftw.irfftn)`` or to real-to-real planner pairs
    (``fftw.dctn / idctn / dstn / idstn``, usually through ``functools.partial(..., type=k)``).
    FFTW-only keywords (planner_effort, threads, overwrite_input) are accepted and ignored.

    ``U`` / ``V``: optionally reuse existing device arrays as the work arrays (PFFT chains the
    stages of a single-GPU transform through shared buffers this way).
    """
    def __init__(self, shape, axes=None, dtype=float, padding=False, backend='fftw',
                 transforms=None, U=None, V=None, **kw):
        FFTBase.__init__(self, shape, axes, dtype, padding)
        if backend not in ('fftw', 'gfft', None):
            raise NotImplementedError("backend %r: this package has a single MI355X engine "
                                      "(no multi-backend dispatch)" % (backend,))
        self.backend = 'gfft'
        transforms = {} if transforms is None else transforms
        if tuple(self.axes) in transforms:
            plan_fwd, plan_bck = transforms[tuple(self.axes)]
        elif self.real_transform:
            plan_fwd, plan_bck = fftw.rfftn, fftw.irfftn
        else:
            plan_fwd, plan_bck = fftw.fftn, fftw.ifftn
        s = tuple(np.take(self.shape, self.axes))
        if U is None:
            U = fftw.aligned(self.shape, dtype=self.dtype)
            U.fill(0)
        assert tuple(U.shape) == tuple(self.shape) and U.dtype == self.dtype
        self.padding_factor = 1.0
        if padding is not False:
            self.padding_factor = padding[self.axes[-1]] if np.ndim(padding) else padding
        self._fused_trunc = False
        if self._padded:
            assert len(self.axes) == 1
            cdtype = np.dtype(self.dtype.char.upper())
            trunc_array = self._get_truncarray(shape, cdtype)
            n_keep = trunc_array.shape[self.axes[-1]]
            # plan against the full-size spectrum's SHAPE only; if the engine fuses truncation and
            # padding into the transform, that array never exists
            full = list(self.shape)
            if self.real_transform:
                full[self.axes[-1]] = full[self.axes[-1]] // 2 + 1
            ghost = _ShapeOnly(full, cdtype)
            self.fwd = plan_fwd(U, s=s, axes=self.axes, output_array=ghost)
            self.bck = plan_bck(ghost, s=s, axes=self.axes, output_array=U)
            if self.fwd.set_truncation(n_keep) and self.bck.set_truncation(n_keep):
                self._fused_trunc = True
            else:
                self.fwd.destroy()
                self.bck.destroy()
                self.fwd = plan_fwd(U, s=s, axes=self.axes, output_array=V)
                V = self.fwd.output_array
                self.bck = plan_bck(V, s=s, axes=self.axes, output_array=U)
            self.M = self.fwd.get_normalization()
            self.forward = _Xfftn_wrap(self._forward, U, trunc_array)
            self.backward = _Xfftn_wrap(self._backward, trunc_array, U)
        else:
            self.fwd = plan_fwd(U, s=s, axes=self.axes, output_array=V)
            V = self.fwd.output_array
            self.bck = plan_bck(V, s=s, axes=self.axes, output_array=U)
            self.M = self.fwd.get_normalization()
            self.forward = _Xfftn_wrap(self._forward, U, V)
            self.backward = _Xfftn_wrap(self._backward, V, U)

    @property
    def _padded(self):
        return abs(self.padding_factor - 1.0) > 1e-8

    def _forward(self, **kw):
        # `src`: read the input directly from a caller's device array of the planned layout
        # (no plan of this package writes to its input when run out of place)
        # `dst`: write the result directly into a caller's device array of the planned layout
        normalize = kw.pop('normalize', True)
        src = kw.pop('src', None)
        dst = kw.pop('dst', None)
        src = self.fwd.input_array if src is None else src
        scale = self.M if normalize else 1.0
        if self._fused_trunc:
            self.fwd.execute_scaled(src, self.forward.output_array if dst is None else dst, scale)
        elif not self._padded:
            self.fwd.execute_scaled(src, self.fwd.output_array if dst is None else dst, scale)
        else:
            self.fwd.execute_scaled(src, self.fwd.output_array, 1.0)
            self._truncation_forward(self.fwd.output_array, self.forward.output_array if dst is None else dst, scale)
        return self.forward.output_array if dst is None else dst

    def _backward(self, **kw):
        normalize = kw.pop('normalize', False)
        src = kw.pop('src', None)
        dst = kw.pop('dst', None)
        out = self.bck.output_array if dst is None else dst
        if self._fused_trunc:
            self.bck.execute_scaled(self.backward.input_array if src is None else src,
                                    out, self.M if normalize else 1.0)
            return out
        if self._padded:
            self._padding_backward(self.backward.input_array if src is None else src, self.bck.input_array)
            src = None
        self.bck.execute_scaled(self.bck.input_array if src is None else src, out,
                                self.M if normalize else 1.0)
        return out

    # 3/2-rule helpers: libfft.py:263-311 as one kernel each
    def _truncation_forward(self, padded_array, trunc_array, scale=1.0):
        axis = self.axes[-1]
        _lib.engine().truncate(padded_array.tensor, trunc_array.tensor, padded_array.shape, axis,
                               trunc_array.shape[axis], self.real_transform,
                               _lib.precision_of(padded_array.dtype), scale)

    def _padding_backward(self, trunc_array, padded_array):
        axis = self.axes[-1]
        _lib.engine().pad(trunc_array.tensor, padded_array.tensor, padded_array.shape, axis,
                          trunc_array.shape[axis], self.real_transform,
                          _lib.precision_of(padded_array.dtype))

    def _get_truncarray(self, shape, dtype):
        axis = self.axes[-1]
        shape = list(shape)
        shape[axis] = int(np.round(shape[axis] / self.padding_factor))
        if self.real_transform:
            shape[axis] = shape[axis] // 2 + 1
        return fftw.aligned(shape, dtype=dtype)

    def destroy(self):
        self.fwd.destroy()
        self.bck.destroy()
