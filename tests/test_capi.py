"""The C-ABI library loads, exports every symbol include/gfft.h declares, and fails loudly
(no host fallback) when no HIP device is present.  No compute calls."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, 'include', 'gfft.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(gfft_[a-z0-9_]+)\s*\(', text)))


def test_header_symbols_exported():
    from mpi4py_fft_amd import _lib
    lib = _lib.lib()
    names = _declared()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), 'libgfft.so does not export %s' % n
    # and the Python binding declares a signature for each of them
    for n in names:
        assert n in _lib.EXPORTS or n in ('gfft_plan_pass_info', 'gfft_plan_profile'), n


def test_error_strings():
    from mpi4py_fft_amd import _lib
    lib = _lib.lib()
    assert lib.gfft_strerror(0) == b'success'
    assert lib.gfft_strerror(-3) == b'no HIP device'
    assert lib.gfft_version() >= 100


def test_bad_arguments_rejected_before_touching_a_device():
    from mpi4py_fft_amd import _lib
    lib = _lib.lib()
    h = ctypes.c_void_p()
    s = (ctypes.c_int64 * 2)(8, 8)
    ax = (ctypes.c_int * 2)(0, 0)
    assert lib.gfft_plan_create(ctypes.byref(h), 2, s, s, 2, ax, -1, 8) == -1       # repeated axis
    ax = (ctypes.c_int * 1)(1)
    assert lib.gfft_plan_create(ctypes.byref(h), 2, s, s, 1, ax, 5, 8) == -2        # r2r kind: unsupported
    assert lib.gfft_plan_create(ctypes.byref(h), 2, s, s, 1, ax, -2, 8) == -1       # r2c shape mismatch
    assert lib.gfft_plan_create(ctypes.byref(h), 2, s, s, 1, ax, -1, 3) == -1       # bad precision


def test_no_host_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip('a GPU is present')
    from mpi4py_fft_amd import _lib, PFFT, comm, fftw
    assert _lib.device_count() == 0
    with pytest.raises(RuntimeError, match='no HIP device'):
        PFFT(comm.COMM_SELF, (8, 8, 8), dtype='D')
    with pytest.raises(RuntimeError):
        fftw.fftn(fftw.aligned((8,), dtype='D'))
    a = fftw.aligned((4, 4), dtype='D')
    with pytest.raises(_lib.GfftError):
        _lib.engine().pack(a.tensor, a.tensor, (4, 4), 0, 2, 16)
