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


# ---- host <-> device staging ----------------------------------------------------------------------
# The reference hands FFTW page-aligned host arrays (fftw/utilities.pyx:54-104, `aligned`); the
# counterpart on this side of PCIe is page-LOCKED host memory.  Measured on the MI355X boxes
# (tools/staging_probe.py, 4 GiB): host -> device from ordinary (pageable) numpy memory already runs
# at the link rate (56.6 GB/s: the runtime DMAs it directly), so `u[...] = host` is a plain copy;
# device -> pageable host is the slow direction (7-8 GB/s).  `np.asarray(u)` therefore lands chunks in
# two pinned bounce buffers at the link rate and drains them into the result with several host
# threads while the next chunk is on the wire; `host_empty` gives the caller a numpy array that IS
# pinned, which moves at 57 GB/s both ways with no staging (`u.get(out=h)`).
PIN_CHUNK_BYTES = 64 << 20
PIN_MIN_BYTES = 32 << 20
HOST_COPY_THREADS = 16
_pinned = {}
_pinned_lock = __import__('threading').Lock()     # the bounce buffers are shared by a process's threads
_pool = None


def _check_async():
    from . import _lib
    _lib.check_async()


def _item(t):
    """A 0-d tensor as a Python scalar: the read synchronises, so this is where a launch that voided itself on the
    device (gfft_async_error) must surface instead of a garbage number."""
    v = t.item()
    if t.is_cuda:
        _check_async()
    return v


def _bounce(device):
    key = str(device)
    b = _pinned.get(key)
    if b is None:
        b = _pinned[key] = dict(buf=[torch.empty(PIN_CHUNK_BYTES, dtype=torch.uint8, pin_memory=True) for _ in range(2)],
                                ev=[torch.cuda.Event(), torch.cuda.Event()])
    return b


def _bytes_view(t):
    return (torch.view_as_real(t) if t.is_complex() else t).reshape(-1).view(torch.uint8)


def _host_copy(dst, src):
    """dst[:] = src for 1-D uint8 numpy arrays, split over a few threads (numpy copies release the
    GIL; one thread moves 6-9 GB/s, far below the link)."""
    global _pool
    n = dst.shape[0]
    k = min(HOST_COPY_THREADS, max(1, n >> 22))
    if k == 1:
        dst[:] = src
        return
    if _pool is None:
        from concurrent.futures import ThreadPoolExecutor
        _pool = ThreadPoolExecutor(HOST_COPY_THREADS)
    step = -(-n // k)
    list(_pool.map(lambda o: dst.__setitem__(slice(o, o + step), src[o:o + step]), range(0, n, step)))


def h2d(dst, host):
    """dst (contiguous device tensor) <- host (C-contiguous numpy array of the same dtype / size)."""
    _bytes_view(dst).copy_(torch.from_numpy(host.reshape(-1).view(np.uint8)))


def d2h(src):
    """numpy copy of a contiguous device tensor (raw bytes; the caller views them)."""
    s = _bytes_view(src)
    n = s.numel()
    out = np.empty(n, dtype=np.uint8)
    if n < PIN_MIN_BYTES or not src.is_cuda:
        torch.from_numpy(out).copy_(s)
        return out
    with _pinned_lock:         # one staged read-back at a time per process (the link is shared anyway)
        b = _bounce(src.device)
        chunk = b['buf'][0].numel()
        pending = None
        for k, off in enumerate(range(0, n, chunk)):
            m = min(chunk, n - off)
            i = k & 1
            b['buf'][i][:m].copy_(s[off:off + m], non_blocking=True)      # buffer i was drained two rounds ago
            b['ev'][i].record()
            if pending is not None:                      # drain the previous chunk while this one flies
                j, poff, pm = pending
                b['ev'][j].synchronize()
                _host_copy(out[poff:poff + pm], b['buf'][j][:pm].numpy())
            pending = (i, off, m)
        j, poff, pm = pending
        b['ev'][j].synchronize()
        _host_copy(out[poff:poff + pm], b['buf'][j][:pm].numpy())
    return out


def host_empty(shape, dtype=float):
    """A numpy array in page-locked host memory: assigning it to a device array, or copying a device
    array into it (``u.get(out=h)``), runs at the PCIe rate with no staging copy."""
    dtype = np.dtype(dtype)
    n = int(np.prod(shape, dtype=np.int64)) * dtype.itemsize
    pin = torch.cuda.is_available()
    raw = torch.empty(max(n, 1), dtype=torch.uint8, pin_memory=pin)
    return raw.numpy()[:n].view(dtype).reshape(shape)


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
    def get(self, out=None):
        """Copy to a host numpy array (a new one, or `out` -- e.g. from `host_empty`)."""
        t = self._t.detach()
        if not t.is_cuda:
            a = t.numpy().copy()
        elif out is not None:
            assert out.shape == self._shape and out.dtype == self._dtype and out.flags.c_contiguous
            torch.from_numpy(out.reshape(-1).view(np.uint8)).copy_(_bytes_view(t.contiguous()))
            _check_async()
            return out
        else:
            a = d2h(t.contiguous()).view(self._dtype).reshape(self._shape)
            _check_async()            # (the copy synchronised: were the launches it waited for valid?)
        if out is not None:
            out[...] = a
            return out
        return a

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

    def _load_host(self, value):
        """Whole-array assignment from a host array through the pinned staging path; False if
        `value` is not a host array of this shape."""
        if isinstance(value, (DeviceArray, torch.Tensor)) or np.isscalar(value) or not self._t.is_cuda:
            return False
        a = np.asarray(value)
        if a.shape != self._shape or not self._t.is_contiguous():
            return False
        a = np.ascontiguousarray(a if a.dtype == self._dtype else a.astype(self._dtype))
        h2d(self._t, a)
        return True

    def set(self, value):
        self[...] = value
        return self

    def __setitem__(self, key, value):
        whole = key is Ellipsis or (isinstance(key, slice) and key == slice(None))
        if whole and self._load_host(value):
            return
        v = self._as_tensor(value)
        if isinstance(key, DeviceArray):
            key = key._t
        if whole:
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
            return _item(sub)
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
        return _item(r) if r.ndim == 0 else self._new(r)

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
