"""The 18 serial `libfft.FFT` fixtures the reference produced (tests/golden/libfft.npz: c2c / r2c,
fp64 / fp32, axes subsets, padding 1.5 / 2.0) fed to the HIP `libfft.FFT` directly -- the same
cases tests/test_oracle_golden.py::test_libfft_cases pins the oracle with."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from tests import cases


def _ncases():
    k = cases.load('libfft')
    i = 0
    while 'libfft%d/A' % i in k.files:
        i += 1
    return i


@pytest.mark.parametrize('i', range(_ncases()))
def test_hip_libfft_matches_reference_fixture(i):
    from mpi4py_fft_amd.libfft import FFT
    k = cases.load('libfft')
    p = 'libfft%d/' % i
    dt = str(k[p + 'dtype'])
    axes = [int(a) for a in k[p + 'axes']]
    axes = None if axes == [-99] else axes
    pad = float(k[p + 'padding'])
    A, B, A2 = k[p + 'A'], k[p + 'B'], k[p + 'A2']
    f = FFT(tuple(int(s) for s in k[p + 'shape']), axes, dt, padding=(pad if pad else False))
    tol = 1e-13 if dt in 'dD' else 2e-5
    got = np.asarray(f.forward(A)).copy()
    assert got.shape == B.shape and got.dtype == B.dtype
    assert np.abs(got - B).max() <= tol * max(1, np.abs(B).max()), np.abs(got - B).max()
    back = np.asarray(f.backward(B)).copy()
    assert back.shape == A2.shape and back.dtype == A2.dtype
    assert np.abs(back - A2).max() <= 20 * tol * max(1, np.abs(A2).max()), np.abs(back - A2).max()
    f.destroy()


def test_fixture_count():
    assert _ncases() >= 18
