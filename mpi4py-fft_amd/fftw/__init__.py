"""Planner API for serial (per-rank) transforms on device memory.

Same surface as mpi4py_fft/fftw (``fftn, ifftn, rfftn, irfftn, get_planned_FFT, FFT, aligned,
aligned_like, get_alignment, get_normalization, fftlib, FFTW_*``), so code written against the
reference's FFTW wrapper runs on the MI355X engine.  A planner function allocates the output
array, computes the ``1/M`` normalisation and returns a plan object (:class:`FFT`) that owns a
``gfft_plan`` handle of libgfft.so; calling the object executes the plan on the bound device
arrays.  The real-to-real planners ``dctn, idctn, dstn, idstn`` (types 1-4, FFTW's unnormalised
REDFTxx / RODFTxx definitions) plan through ``gfft_plan_create_r2r``; ``hfftn / ihfftn`` are the
c2r / r2c plans under their other names, as in the reference.  The halfcomplex / Hartley
kinds raise ``NotImplementedError``; wisdom and time limits are accepted and have nothing to do
(plans are deterministic).
"""
import numpy as np

from .. import _lib
from ..array import DeviceArray, empty as _empty

# constants of mpi4py_fft/fftw/utilities.pyx:7-37 (values are FFTW's)
FFTW_FORWARD, FFTW_BACKWARD = -1, 1
FFTW_R2HC, FFTW_HC2R, FFTW_DHT = 0, 1, 2
FFTW_REDFT00, FFTW_REDFT01, FFTW_REDFT10, FFTW_REDFT11 = 3, 4, 5, 6
FFTW_RODFT00, FFTW_RODFT01, FFTW_RODFT10, FFTW_RODFT11 = 7, 8, 9, 10
C2C_FORWARD, C2C_BACKWARD, R2C, C2R = -1, 1, -2, 2
FFTW_MEASURE, FFTW_DESTROY_INPUT, FFTW_UNALIGNED, FFTW_CONSERVE_MEMORY = 0, 1, 2, 4
FFTW_EXHAUSTIVE, FFTW_PRESERVE_INPUT, FFTW_PATIENT, FFTW_ESTIMATE = 8, 16, 32, 64
FFTW_WISDOM_ONLY = 2097152

flag_dict = {k: v for k, v in dict(locals()).items() if k.startswith('FFTW_')}


def get_alignment(array):
    """Largest power of two <= 32 dividing the buffer address (utilities.pyx:39-52)."""
    addr = array.data_ptr if isinstance(array, DeviceArray) else array.ctypes.data
    for i in range(5, -1, -1):
        if addr % (1 << i) == 0:
            return 1 << i
    return 1


def aligned(shape, n=32, dtype=np.dtype('d'), fill=None):
    """Device array of `shape` (allocations are >= 256-byte aligned, so any n <= 32 holds)."""
    a = _empty(shape, np.dtype(dtype))
    if fill is not None:
        assert isinstance(fill, int)
        a.fill(fill)
    return a


def aligned_like(z, fill=None):
    return aligned(z.shape, n=get_alignment(z), dtype=z.dtype, fill=fill)


class FFT:
    """Plan object: the device counterpart of the Cython class ``fftw_xfftn.FFT``
    (fftw_xfftn.pyx:50-296).  Holds a ``gfft_plan`` plus the arrays it was planned for."""
    def __init__(self, input_array, output_array, axes=(-1,), kind=FFTW_FORWARD, threads=1,
                 flags=FFTW_MEASURE, normalization=1.0):
        nd = len(input_array.shape)
        self.axes = tuple(a + nd if a < 0 else a for a in axes)
        kinds = [int(k) for k in kind] if isinstance(kind, (list, tuple, np.ndarray)) else [int(kind)]
        r2r = all(FFTW_REDFT00 <= k <= FFTW_RODFT11 for k in kinds)
        if r2r:
            # one kind per axis (fftw_planxfftn.c:68-75)
            kinds = kinds * len(self.axes) if len(kinds) == 1 else kinds
            assert len(kinds) == len(self.axes)
            assert tuple(input_array.shape) == tuple(output_array.shape)
        else:
            kind = kinds[0]
            if kind not in (C2C_FORWARD, C2C_BACKWARD, R2C, C2R):
                raise NotImplementedError('halfcomplex / Hartley kinds (R2HC, HC2R, DHT) are not implemented')
        self.kind = tuple(kinds) if r2r else kind
        self._M = float(normalization)
        self._input_array = input_array
        self._output_array = output_array
        self.input_shape, self.output_shape = tuple(input_array.shape), tuple(output_array.shape)
        self.input_strides, self.output_strides = input_array.strides, output_array.strides
        self._precision = _lib.precision_of(input_array.dtype)
        self._eng = _lib.engine()
        try:
            if r2r:
                self._plan = self._eng.plan_create_r2r(self.input_shape, self.axes, self.kind, self._precision)
            else:
                self._plan = self._eng.plan_create(self.input_shape, self.output_shape, self.axes,
                                                   kind, self._precision)
        except _lib.GfftError as e:
            # same failure mode as fftw_xfftn.pyx:152-153
            raise RuntimeError('Failure creating gfft plan: %s' % e)

    @classmethod
    def padded(cls, input_array, output_array, kind, normalization=1.0):
        """The whole padded 3-D transform of a one-rank PFFT as one plan (gfft_plan_create_padded):
        the physical side has the padded shape, the spectral side the truncated one.  None when the
        engine keeps the per-axis form."""
        eng = _lib.engine()
        if not hasattr(eng, 'plan_create_padded'):
            return None
        inverse = kind in (C2C_BACKWARD, C2R)
        phys, spec = (output_array, input_array) if inverse else (input_array, output_array)
        h = eng.plan_create_padded(phys.shape, spec.shape, kind, _lib.precision_of(input_array.dtype))
        if h is None:
            return None
        self = cls.__new__(cls)
        self.axes, self.kind, self._M = (0, 1, 2), kind, float(normalization)
        self._input_array, self._output_array = input_array, output_array
        self.input_shape, self.output_shape = tuple(input_array.shape), tuple(output_array.shape)
        self.input_strides, self.output_strides = input_array.strides, output_array.strides
        self._precision, self._eng, self._plan = _lib.precision_of(input_array.dtype), eng, h
        return self

    def __del__(self):
        self.destroy()

    def destroy(self):
        plan, self._plan = getattr(self, '_plan', None), None
        if plan is not None:
            try:
                self._eng.plan_destroy(plan)
            except Exception:
                pass

    input_array = property(lambda self: self._input_array)
    output_array = property(lambda self: self._output_array)

    def print_plan(self):
        print(self._eng.plan_describe(self._plan))

    def cost(self):
        """(flops, algorithmic bytes, kernel launches) of one execution."""
        return self._eng.plan_cost(self._plan)

    def profile(self):
        """Per-pass (family, algorithmic bytes/launch, total ms, launches) since the last call."""
        return self._eng.plan_profile(self._plan, self.cost()[2])

    def set_truncation(self, n_keep):
        """Fuse the 3/2-rule truncation (forward kinds) / zero padding (backward kinds) into this
        single-axis plan; returns False when the engine cannot (see gfft_plan_set_truncation)."""
        return self._eng.plan_set_truncation(self._plan, n_keep)

    def set_split(self, side, nblocks):
        """Read the input from (side 0) / write the output as (side 1) an all-to-all buffer of
        `nblocks` equal blocks of the transformed axis (see gfft_plan_set_split); False when the
        engine cannot fuse it into this plan."""
        return self._eng.plan_set_split(self._plan, side, nblocks)

    def update_arrays(self, input_array, output_array):
        assert self.input_shape == tuple(input_array.shape)
        assert self.input_strides == input_array.strides
        assert self._input_array.dtype == input_array.dtype
        assert self.output_shape == tuple(output_array.shape)
        assert self.output_strides == output_array.strides
        assert self._output_array.dtype == output_array.dtype
        self._input_array, self._output_array = input_array, output_array

    def get_normalization(self):
        return self._M

    def _compatible(self, a, shape, strides, dtype):
        return (isinstance(a, DeviceArray) and tuple(a.shape) == shape and a.strides == strides
                and a.dtype == dtype and a.is_contiguous())

    def __call__(self, input_array=None, output_array=None, implicit=True, normalize=False, **kw):
        """Execute.  implicit=True runs directly on the given arrays when they match the planned
        layout, otherwise (or with implicit=False) the input is first copied into the plan's own
        array -- the same contract as fftw_xfftn.pyx:195-296.  ``normalize`` multiplies by 1/M,
        fused into the last kernel."""
        tin = self._input_array
        if input_array is not None:
            if implicit and self._compatible(input_array, self.input_shape, self.input_strides,
                                             self._input_array.dtype):
                tin = input_array
            else:
                self._input_array[...] = input_array
        tout = self._output_array
        copy_out = None
        if output_array is not None:
            if implicit:
                assert self._compatible(output_array, self.output_shape, self.output_strides,
                                        self._output_array.dtype), 'output_array has wrong layout'
                tout = output_array
            else:
                copy_out = output_array
        self._eng.plan_execute(self._plan, tin.tensor, tout.tensor, self._M if normalize else 1.0)
        if copy_out is not None:
            copy_out[...] = tout
            return copy_out
        return tout

    def status(self):
        """After synchronising: raise RuntimeError if a launch of this plan voided itself on the device (a fused pass
        pair that waited too long for another workgroup, include/gfft.h gfft_plan_status); the plan's last results are
        then invalid and it runs stand-alone passes from now on."""
        if hasattr(self._eng, 'plan_status'):
            self._eng.plan_status(self._plan)

    def execute_scaled(self, tin, tout, scale):
        """Internal: run on explicit arrays with an arbitrary fused scale factor."""
        self._eng.plan_execute(self._plan, tin.tensor, tout.tensor, scale)
        return tout


def get_planned_FFT(input_array, output_array, axes=(-1,), kind=FFTW_FORWARD, threads=1,
                    flags=(FFTW_MEASURE,), normalization=1.0):
    assert input_array.dtype.char.upper() in fftlib, 'long double has no GPU type'
    _check_in(input_array)
    return FFT(input_array, output_array, axes, kind, threads, flags, normalization)


def _check_in(a):
    assert isinstance(a, DeviceArray) or all(hasattr(a, k) for k in ('shape', 'dtype', 'strides')), \
        'planner functions take device arrays (see fftw.aligned)'


def fftn(input_array, s=None, axes=(-1,), threads=1, flags=(FFTW_MEASURE,), output_array=None):
    """Plan a complex-to-complex forward transform (xfftn.py:38-104)."""
    _check_in(input_array)
    assert input_array.dtype.char in 'FD'
    if output_array is None:
        output_array = aligned(input_array.shape, dtype=input_array.dtype)
    else:
        assert tuple(input_array.shape) == tuple(output_array.shape)
        assert output_array.dtype.char == input_array.dtype.char.upper()
    M = np.prod(np.take(input_array.shape, axes))
    return get_planned_FFT(input_array, output_array, axes, FFTW_FORWARD, threads, flags, 1.0 / M)


def ifftn(input_array, s=None, axes=(-1,), threads=1, flags=(FFTW_MEASURE,), output_array=None):
    """Plan a complex-to-complex backward (unnormalised inverse) transform (xfftn.py:106-171)."""
    _check_in(input_array)
    assert input_array.dtype.char in 'FD'
    if output_array is None:
        output_array = aligned_like(input_array)
    else:
        assert tuple(input_array.shape) == tuple(output_array.shape)
    M = np.prod(np.take(input_array.shape, axes))
    return get_planned_FFT(input_array, output_array, axes, FFTW_BACKWARD, threads, flags, 1.0 / M)


def rfftn(input_array, s=None, axes=(-1,), threads=1, flags=(FFTW_MEASURE,), output_array=None):
    """Plan a real-to-complex transform; the halved axis is axes[-1] (xfftn.py:173-240)."""
    _check_in(input_array)
    assert input_array.dtype.char in 'fd'
    if output_array is None:
        sz = list(input_array.shape)
        sz[axes[-1]] = input_array.shape[axes[-1]] // 2 + 1
        output_array = aligned(sz, dtype=np.dtype(input_array.dtype.char.upper()))
    else:
        assert input_array.shape[axes[-1]] // 2 + 1 == output_array.shape[axes[-1]]
    M = np.prod(np.take(input_array.shape, axes))
    return get_planned_FFT(input_array, output_array, axes, R2C, threads, flags, 1.0 / M)


def irfftn(input_array, s=None, axes=(-1,), threads=1, flags=(FFTW_MEASURE,), output_array=None):
    """Plan a complex-to-real transform; without `s` (or output_array) the real length along
    axes[-1] is assumed even, 2(n-1) (xfftn.py:242-326)."""
    _check_in(input_array)
    assert input_array.dtype.char in 'FD'
    assert FFTW_PRESERVE_INPUT not in flags
    sz = list(input_array.shape)
    if s is not None:
        assert len(axes) == len(s)
        for q, axis in zip(s, axes):
            sz[axis] = q
    elif output_array is not None:
        sz = list(output_array.shape)
    else:
        sz[axes[-1]] = 2 * sz[axes[-1]] - 2
    if output_array is None:
        output_array = aligned(sz, dtype=np.dtype(input_array.dtype.char.lower()))
    else:
        assert list(output_array.shape) == sz
    assert sz[axes[-1]] // 2 + 1 == input_array.shape[axes[-1]]
    M = np.prod(np.take(output_array.shape, axes))
    return get_planned_FFT(input_array, output_array, axes, C2R, threads, flags, 1.0 / M)


# real-to-real planners (xfftn.py:14-36,328-614): `type` selects the FFTW kind per the tables below
dct_type = {1: FFTW_REDFT00, 2: FFTW_REDFT10, 3: FFTW_REDFT01, 4: FFTW_REDFT11}
idct_type = {1: FFTW_REDFT00, 2: FFTW_REDFT01, 3: FFTW_REDFT10, 4: FFTW_REDFT11}
dst_type = {1: FFTW_RODFT00, 2: FFTW_RODFT10, 3: FFTW_RODFT01, 4: FFTW_RODFT11}
idst_type = {1: FFTW_RODFT00, 2: FFTW_RODFT01, 3: FFTW_RODFT10, 4: FFTW_RODFT11}


def _r2r_planner(name, table):
    def plan(input_array, s=None, axes=(-1,), type=2, threads=1, flags=(FFTW_MEASURE,), output_array=None):
        _check_in(input_array)
        assert input_array.dtype.char in 'fd'
        if output_array is None:
            output_array = aligned_like(input_array)
        else:
            assert tuple(input_array.shape) == tuple(output_array.shape)
        kind = [table[type]] * len(axes)
        M = get_normalization(kind, input_array.shape, axes)
        return get_planned_FFT(input_array, output_array, axes, kind, threads, flags, M)
    plan.__name__ = name
    plan.__doc__ = ('Plan a real-to-real %s over `axes`; `type` 1-4 -> %s (xfftn.py:328-614).  '
                    'Unnormalised FFTW definitions; `s` is unused.' % (name, sorted(table.items())))
    return plan


dctn, idctn = _r2r_planner('dctn', dct_type), _r2r_planner('idctn', idct_type)
dstn, idstn = _r2r_planner('dstn', dst_type), _r2r_planner('idstn', idst_type)


def hfftn(input_array, s=None, axes=(-1,), threads=1, flags=(FFTW_MEASURE,), output_array=None):
    """Plan the transform of an array with Hermitian symmetry (xfftn.py:684-761): the same
    complex-to-real plan `irfftn` builds (kind C2R), without its PRESERVE_INPUT restriction."""
    return irfftn(input_array, s, axes, threads, tuple(f for f in flags if f != FFTW_PRESERVE_INPUT), output_array)


def ihfftn(input_array, s=None, axes=(-1,), threads=1, flags=(FFTW_MEASURE,), output_array=None):
    """Plan the inverse of `hfftn` (xfftn.py:616-682): the real-to-complex plan of `rfftn`."""
    return rfftn(input_array, s, axes, threads, flags, output_array)


def get_normalization(kind, shape, axes):
    """1 / product of the logical lengths of the transformed axes (xfftn.py:763-816): N for the
    Fourier kinds, 2(N-1) for REDFT00, 2(N+1) for RODFT00, 2N for the other real-to-real kinds."""
    kind = [kind] * len(axes) if isinstance(kind, (int, np.integer)) else kind
    assert len(kind) == len(axes)
    M = 1
    for knd, axis in zip(kind, axes):
        N = shape[axis]
        if knd == FFTW_RODFT00:
            M *= 2 * (N + 1)
        elif knd == FFTW_REDFT00:
            M *= 2 * (N - 1)
        elif knd in (FFTW_RODFT01, FFTW_RODFT10, FFTW_RODFT11, FFTW_REDFT01, FFTW_REDFT10, FFTW_REDFT11):
            M *= 2 * N
        elif knd in (FFTW_FORWARD, FFTW_BACKWARD, R2C, C2R):
            M *= N
        else:
            raise NotImplementedError('halfcomplex / Hartley kinds are not implemented')
    return 1. / M


inverse = {FFTW_RODFT11: FFTW_RODFT11, FFTW_REDFT11: FFTW_REDFT11, FFTW_RODFT01: FFTW_RODFT10,
           FFTW_RODFT10: FFTW_RODFT01, FFTW_REDFT01: FFTW_REDFT10, FFTW_REDFT10: FFTW_REDFT01,
           FFTW_RODFT00: FFTW_RODFT00, FFTW_REDFT00: FFTW_REDFT00,
           rfftn: irfftn, irfftn: rfftn, fftn: ifftn, ifftn: fftn}


class _PrecisionLib:
    """What ``fftlib['D']`` / ``fftlib['F']`` expose in the reference (factory.py:44-48)."""
    def __init__(self, char):
        self.char = char
        self.FFT = FFT


fftlib = {'F': _PrecisionLib('F'), 'D': _PrecisionLib('D')}


def get_fftw_lib(dtype):
    """The engine for a precision, or None (long double has no GPU type) (factory.py:7-42)."""
    return fftlib.get(np.dtype(dtype).char.upper())


def export_wisdom(filename):
    """FFTW wisdom is accumulated planner measurements (utilities.pyx / fftw_xfftn.pyx:298-330).
    Plans here are deterministic functions of shape, axes and kind, so there is nothing to save:
    the call leaves an empty file so that a later ``import_wisdom`` finds one."""
    open(filename, 'a').close()


def import_wisdom(filename):
    open(filename, 'rb').close()


def forget_wisdom():
    pass


def set_timelimit(limit):
    """Planning never searches, so no time limit applies."""


def cleanup():
    pass
