"""Physics known-answer test reachable through the path (examples/spectral_dns_solver.py:129 of the
reference): Taylor-Green vortex, 64^3, RK4 x 10 steps -> kinetic energy 0.124953117517 to 7
decimals.  Runs the device port of that example on 1 rank and on 2 / 4 thread-ranks."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'examples'))


@pytest.mark.parametrize('P', [1, 2, 4])
def test_taylor_green_energy(P):
    from dns_taylor_green import solve
    from tests import cases
    energies = cases.run_ranks(P, lambda comm: solve(comm))
    for e in energies:
        assert round(e - 0.124953117517, 7) == 0, e


@pytest.mark.parametrize('P', [1, 2])
def test_taylor_green_energy_torch_expression_path(P):
    """The line-by-line transcription of the reference's expressions (A/B baseline of the example)."""
    from dns_taylor_green import solve
    from tests import cases
    for e in cases.run_ranks(P, lambda comm: solve(comm, fused=False)):
        assert round(e - 0.124953117517, 7) == 0, e


@pytest.mark.parametrize('dt', ['d', 'f'])
def test_spectral_kernels_against_expressions(dt):
    """csrc/spectral.hip vs the numpy expressions of examples/spectral_dns_solver.py:65-91 on the
    same data (rounding-level agreement: the operations are reassociated, not approximated)."""
    import numpy as np
    import torch
    from mpi4py_fft_amd import PFFT, comm, newDistArray, spectral
    shape, L, nu = (24, 16, 20), np.array([2 * np.pi, 4 * np.pi, 6 * np.pi]), 0.03
    fft = PFFT(comm.COMM_SELF, shape, dtype=dt)
    ops = spectral.SpectralOps(fft, L)
    rng = np.random.default_rng(5)
    cdt = np.dtype(dt.upper())
    tol = 1e-13 if dt == 'd' else 2e-5
    # wavenumber mesh exactly as the reference builds it
    k = [np.fft.fftfreq(n, 1. / n).astype(int) for n in shape[:-1]] + [np.fft.rfftfreq(shape[-1], 1. / shape[-1]).astype(int)]
    Ks = np.meshgrid(*k, indexing='ij', sparse=True)
    K = np.array([np.broadcast_to(kk * (2 * np.pi / L[i]), fft.shape(True)) for i, kk in enumerate(Ks)]).astype(float)
    K2 = np.sum(K * K, 0)
    K_over_K2 = K / np.where(K2 == 0, 1, K2)
    for kk, ref in zip(ops.K, (K[0][:, 0, 0], K[1][0, :, 0], K[2][0, 0, :])):
        assert np.allclose(kk.cpu().numpy(), ref, rtol=1e-6 if dt == 'f' else 1e-15)
    uh = (rng.standard_normal((3,) + fft.shape(True)) + 1j * rng.standard_normal((3,) + fft.shape(True))).astype(cdt)
    du = (rng.standard_normal(uh.shape) + 1j * rng.standard_normal(uh.shape)).astype(cdt)
    U_hat, dU, W = (newDistArray(fft, rank=1) for _ in range(3))
    U_hat[...] = uh
    dU[...] = du
    ops.curl(U_hat, W)
    want = np.array([1j * (K[1] * uh[2] - K[2] * uh[1]), 1j * (K[2] * uh[0] - K[0] * uh[2]), 1j * (K[0] * uh[1] - K[1] * uh[0])])
    assert np.abs(np.asarray(W) - want).max() <= tol * np.abs(want).max()
    ops.project(dU, U_hat, nu)
    rhs = du.astype('D')
    P_hat = np.sum(rhs * K_over_K2, 0)
    rhs = rhs - P_hat * K
    rhs = rhs - nu * K2 * uh
    assert np.abs(np.asarray(dU) - rhs).max() <= tol * np.abs(rhs).max()
    a = rng.standard_normal((3,) + fft.shape(False)).astype(dt)
    b = rng.standard_normal(a.shape).astype(dt)
    A, B, C = (newDistArray(fft, False, rank=1) for _ in range(3))
    A[...] = a
    B[...] = b
    spectral.cross(A, B, C)
    assert np.abs(np.asarray(C) - np.cross(a, b, axis=0)).max() <= tol * 10
    u0 = (rng.standard_normal(uh.shape) + 1j * rng.standard_normal(uh.shape)).astype(cdt)
    u1 = (rng.standard_normal(uh.shape) + 1j * rng.standard_normal(uh.shape)).astype(cdt)
    U0, U1, Un = (newDistArray(fft, rank=1) for _ in range(3))
    U0[...] = u0
    U1[...] = u1
    spectral.rk_stage(Un, U0, U1, dU, 0.25, 0.125)
    assert np.abs(np.asarray(Un) - (u0 + 0.25 * np.asarray(dU))).max() <= tol * 10
    assert np.abs(np.asarray(U1) - (u1 + 0.125 * np.asarray(dU))).max() <= tol * 10
    keep = np.asarray(Un).copy()
    spectral.rk_stage(None, None, U1, dU, 0.0, 0.5)
    assert np.array_equal(np.asarray(Un), keep)
    fft.destroy()


def test_taylor_green_step_replayed_from_a_hip_graph():
    """Every kernel of the package is enqueued on torch's current stream and nothing allocates or
    synchronises after the first execution, so a whole RK4 step (~130 launches at 64^3) can be
    captured into a HIP graph and replayed; the known answer must not change."""
    from dns_taylor_green import solve
    from mpi4py_fft_amd import comm
    e = solve(comm.COMM_SELF, graph=True)
    assert round(e - 0.124953117517, 7) == 0, e
