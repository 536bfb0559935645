"""Device-resident n-d array used wherever the reference uses a numpy array on the hot path.

The reference's ``DistArray`` *is* an ``np.ndarray`` (mpi4py_fft/distarray.py:10) and its FFT
objects own numpy work arrays (libfft.py:72-78).  Here the bytes live in HBM: the storage is a
contiguous ``torch`` tensor (torch is the allocator/stream plumbing), and this class gives it the
small slice of ndarray behaviour the reference's API and tests rely on: ``shape/dtype/ndim/size``,
``a[...] = x`` from numpy / scalars / other device arrays, ``np.asarray(a)`` (device-to-host copy,
so ``np.allclose(a, b)`` and ``np.linalg.norm(a)`` work), ``copy()``, ``fill()``, in-place
arithmetic and basic indexing.  On a machine without a GPU the storage falls back to host memory
so that planning / geometry code can be unit-tested; every compute entry point of the engine
refuses host tensors (``_lib.HipEngine.require_device``).
"""
import numpy as np
import torch

_NP2T = {'f': torch.float32, 'd': torch.float64, 'F': torch.complex64, 'D': torch.complex128,
         'i': torch.int32, 'l': torch.int64, 'b': torch.int8, 'B': torch.uint8}


def default_device():
    if torch.cuda.is_available():
        return torch.device('cuda', torch.cuda.current_device())
    return torch.device('cpu')


def torch_dtype(dtype):
    ch = np.dtype(dtype).char
    if ch not in _NP2T:
        raise TypeError('unsupported dtype %r (supported: f d F D; long double has no GPU type)' % (dtype,))
    return _NP2T[ch]


class DeviceArray:
    __array_priority__ = 100.0
    __array_ufunc__ = None      # numpy defers `ndarray (op) DeviceArray` to the reflected methods below

    def __init__(self, shape, dtype=float, device=None, tensor=None, val=None):
        if np.ndim(shape) == 0:
            shape = (int(shape),)
        self._shape = tuple(int(s) for s in shape)
        self._dtype = np.dtype(dtype)
        if tensor is None:
            dev = default_device() if device is None else torch.device(device)
            tensor = torch.empty(self._shape, dtype=torch_dtype(self._dtype), device=dev)
            if val is not None:
                tensor.fill_(val)
        else:
            assert tuple(tensor.shape) == self._shape, (tuple(tensor.shape), self._shape)
            assert tensor.dtype == torch_dtype(self._dtype)
        self._t = tensor

    # ---- ndarray-like metadata
    shape = property(lambda self: self._shape)
    dtype = property(lambda self: self._dtype)
    ndim = property(lambda self: len(self._shape))
    size = property(lambda self: int(np.prod(self._shape, dtype=np.int64)))
    itemsize = property(lambda self: self._dtype.itemsize)
    nbytes = property(lambda self: self.size * self.itemsize)
    tensor = property(lambda self: self._t)
    device = property(lambda self: self._t.device)

    @property
    def strides(self):
        return tuple(int(s) * self.itemsize for s in self._t.stride())

    @property
    def data_ptr(self):
        return self._t.data_ptr()

    def is_contiguous(self):
        return self._t.is_contiguous()

    def __len__(self):
        return self._shape[0]

    def __repr__(self):
        return '%s(shape=%s, dtype=%s, device=%s)' % (type(self).__name__, self._shape, self._dtype, self._t.device)

    # ---- host interop
    def get(self):
        """Copy to a new host numpy array."""
        t = self._t.detach()
        return t.cpu().numpy() if t.is_cuda else t.numpy().copy()

    def __array__(self, dtype=None, copy=None):
        a = self.get()              # always fresh memory, so `copy=True` is honoured
        return a if dtype is None else a.astype(dtype, copy=False)

    def _as_tensor(self, value):
        if isinstance(value, DeviceArray):
            return value._t
        if isinstance(value, torch.Tensor):
            return value
        if np.isscalar(value):
            return value
        a = np.asarray(value)
        if a.dtype != self._dtype:
            a = a.astype(self._dtype)
        return torch.from_numpy(np.ascontiguousarray(a)).to(self._t.device, non_blocking=False)

    def set(self, value):
        self[...] = value
        return self

    def __setitem__(self, key, value):
        v = self._as_tensor(value)
        if isinstance(key, DeviceArray):
            key = key._t
        if key is Ellipsis or (isinstance(key, slice) and key == slice(None)):
            if isinstance(v, torch.Tensor):
                self._t.copy_(v if v.dtype == self._t.dtype else v.to(self._t.dtype))
            else:
                self._t.fill_(v)
        else:
            self._t[key] = v

    def __getitem__(self, key):
        if isinstance(key, DeviceArray):
            key = key._t
        sub = self._t[key]
        return self._view(sub)

    def _view(self, sub):
        if sub.ndim == 0:
            return sub.item()
        out = DeviceArray.__new__(DeviceArray)
        out._shape = tuple(sub.shape)
        out._dtype = self._dtype
        out._t = sub
        return out

    def fill(self, val):
        self._t.fill_(val)

    def copy(self):
        out = type(self).__new__(type(self))
        out.__dict__.update(self.__dict__)
        out._t = self._t.clone()
        return out

    def astype(self, dtype):
        return DeviceArray(self._shape, dtype, tensor=self._t.to(torch_dtype(dtype)))

    # ---- in-place arithmetic (convenience; the hot path fuses its scaling into kernels)
    def _bin(self, other):
        return other._t if isinstance(other, DeviceArray) else self._as_tensor(other)

    def __imul__(self, other):
        self._t.mul_(self._bin(other))
        return self

    def __iadd__(self, other):
        self._t.add_(self._bin(other))
        return self

    def __isub__(self, other):
        self._t.sub_(self._bin(other))
        return self

    def __itruediv__(self, other):
        self._t.div_(self._bin(other))
        return self

    def _new(self, t):
        inv = {v: k for k, v in _NP2T.items()}
        return DeviceArray(tuple(t.shape), np.dtype(inv[t.dtype]), tensor=t.contiguous())

    def __mul__(self, other):
        return self._new(self._t * self._bin(other))

    __rmul__ = __mul__

    def __add__(self, other):
        return self._new(self._t + self._bin(other))

    __radd__ = __add__

    def __sub__(self, other):
        return self._new(self._t - self._bin(other))

    def __truediv__(self, other):
        return self._new(self._t / self._bin(other))

    def __rsub__(self, other):
        return self._new(self._bin(other) - self._t)

    def sum(self, axis=None, dtype=None, out=None, **kw):
        # signature numpy's np.sum(a) dispatches to for non-ndarray objects
        assert out is None
        r = self._t.sum() if axis is None else self._t.sum(dim=axis)
        return r.item() if r.ndim == 0 else self._new(r)

    def __neg__(self):
        return self._new(-self._t)

    @property
    def real(self):
        return self._new(self._t.real.clone()) if self._dtype.kind == 'c' else self

    @property
    def imag(self):
        return self._new(self._t.imag.clone())

    def conj(self):
        return self._new(self._t.conj().resolve_conj())


def empty(shape, dtype=float, device=None):
    return DeviceArray(shape, dtype, device)


def zeros(shape, dtype=float, device=None):
    return DeviceArray(shape, dtype, device, val=0)


def asdevice(a, dtype=None, device=None):
    """numpy / DeviceArray -> DeviceArray (copies host data to the device)."""
    if isinstance(a, DeviceArray) and (dtype is None or np.dtype(dtype) == a.dtype):
        return a
    h = np.asarray(a)
    if dtype is not None:
        h = h.astype(dtype, copy=False)
    out = DeviceArray(h.shape, h.dtype, device)
    out[...] = h
    return out
