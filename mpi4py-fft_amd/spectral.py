"""The pseudo-spectral caller either side of the transform path.

The reference's demo solver (examples/spectral_dns_solver.py:65-91) writes the steps between a
``backward`` and the next ``forward`` as numpy expressions over array-sized wavenumber meshes.  On
the device each of those steps is one HIP kernel (csrc/spectral.hip) that reads its operands once:

    curl(u_hat, out)            out = 1j * (K x u_hat)                     compute_curl, :76-80
    cross(a, b, out)            out = a x b  (physical space)              cross, :69-74
    project(du_hat, u_hat, nu)  P = sum(du*K/|K|^2); du -= P*K; du -= nu*|K|^2*u_hat      :88-90
    rk_stage(u, u0, u1, du, cb, ca)   u = u0 + cb*du;  u1 += ca*du         :112-116

Fields are ``newDistArray(fft, rank=1)`` arrays ([3][local shape]); the wavenumbers are three
per-axis device vectors (the sparse form of get_local_wavenumbermesh, :52-63).
"""
import numpy as np
import torch

from . import _lib
from .array import DeviceArray


def local_wavenumbers(fft, L=None):
    """Per-axis wavenumber vectors of this rank's block of the spectral array, scaled by
    2 pi / L (examples/spectral_dns_solver.py:52-63): [k0, k1, k2] as 1-D device arrays."""
    s = fft.local_slice(True)
    N = fft.global_shape()
    real = np.dtype(fft.dtype(False)).kind == 'f'
    k = [np.fft.fftfreq(n, 1. / n) for n in N]
    if real:
        k[-1] = np.fft.rfftfreq(N[-1], 1. / N[-1])
    L = np.full(len(N), 2 * np.pi) if L is None else np.asarray(L, dtype=float)
    rdt = np.dtype(fft.dtype(True).char.lower())
    dev = fft.forward.output_array.device
    return [torch.as_tensor((ki[si].astype(int) * (2 * np.pi / L[i])).astype(rdt), device=dev)
            for i, (ki, si) in enumerate(zip(k, s))]


def _prec(a):
    return _lib.precision_of(a.dtype)


def _t(a):
    return a.tensor if isinstance(a, DeviceArray) else a


class SpectralOps:
    """The four kernels bound to one PFFT's local spectral shape and wavenumbers."""
    def __init__(self, fft, L=None):
        assert len(fft.global_shape()) == 3, 'vector calculus kernels are 3-D'
        self.K = local_wavenumbers(fft, L)
        self.shape = tuple(int(n) for n in fft.shape(True))

    def curl(self, u_hat, out):
        """out = 1j * (K x u_hat); both [3] + spectral shape, complex."""
        assert tuple(u_hat.shape) == (3,) + self.shape == tuple(out.shape)
        _lib.engine().ps_curl(_t(u_hat), _t(out), self.K, self.shape, _prec(u_hat))
        return out

    def project(self, du_hat, u_hat, nu):
        """In place: pressure projection and viscous term of the Navier-Stokes right-hand side."""
        assert tuple(du_hat.shape) == (3,) + self.shape == tuple(u_hat.shape)
        _lib.engine().ps_project(_t(du_hat), _t(u_hat), self.K, self.shape, nu, _prec(du_hat))
        return du_hat


def cross(a, b, out):
    """out = a x b for real fields of shape [3] + local physical shape."""
    assert a.shape[0] == 3 and tuple(a.shape) == tuple(b.shape) == tuple(out.shape)
    assert np.dtype(a.dtype).kind == 'f'
    count = int(np.prod(a.shape[1:], dtype=np.int64))
    _lib.engine().ps_cross(_t(a), _t(b), _t(out), count, _prec(a))
    return out


def rk_stage(u, u0, u1, du, cb, ca):
    """u = u0 + cb*du (skipped when u is None); u1 += ca*du -- one pass over the four arrays."""
    mult = 2 if np.dtype(du.dtype).kind == 'c' else 1
    count = int(np.prod(du.shape, dtype=np.int64)) * mult
    _lib.engine().ps_rk_stage(None if u is None else _t(u), None if u0 is None else _t(u0), _t(u1), _t(du),
                              count, cb, ca, _prec(du))
