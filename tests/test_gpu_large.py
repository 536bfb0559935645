"""Size-independent properties at (near) BASELINE sizes, plus oracle comparisons at sizes the
oracle finishes in seconds."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import pfft_oracle as O


def test_256cubed_vs_oracle():
    from mpi4py_fft_amd import PFFT, newDistArray, comm
    shape = (256, 256, 256)
    fft = PFFT(comm.COMM_SELF, shape, dtype='D')
    G = O.rng_array(shape, 'D', 1234)
    u = newDistArray(fft, False)
    u[...] = G
    uh = np.asarray(fft.forward(u))
    import scipy.fft
    ref = scipy.fft.fftn(G, workers=-1) / G.size
    err = np.abs(uh - ref).max() / np.abs(ref).max()
    assert err <= 2e-10 and err < 1e-13, err
    back = np.asarray(fft.backward())
    rt = np.linalg.norm(back - G) / np.linalg.norm(G)
    assert rt <= 1e-10 and rt < 1e-14, rt
    fft.destroy()


@pytest.mark.parametrize('n,dt', [(512, 'D'), (512, 'd'), (1024, 'F')])
def test_cubed_properties(n, dt):
    """Round trip (the north-star tolerance), Parseval and linearity at n^3."""
    import torch
    from mpi4py_fft_amd import PFFT, newDistArray, comm
    shape = (n, n, n)
    fft = PFFT(comm.COMM_SELF, shape, dtype=dt)
    u = newDistArray(fft, False)
    g = torch.Generator(device='cuda').manual_seed(1234)
    u.tensor.copy_(torch.randn(u.tensor.shape, generator=g, device='cuda', dtype=u.tensor.dtype))
    u0 = u.tensor.clone()
    uh = fft.forward(u)
    # Parseval: sum|u|^2 / N == sum|uh|^2 (forward carries 1/N); half spectrum for r2c
    e_phys = float((u0.abs() ** 2).sum().item()) / u0.numel()
    if dt in 'DF':
        e_spec = float((uh.tensor.abs() ** 2).sum().item())
    else:
        w = torch.full((uh.shape[-1],), 2.0, device='cuda', dtype=torch.float64)
        w[0] = 1.0
        if n % 2 == 0:
            w[-1] = 1.0
        e_spec = float(((uh.tensor.abs() ** 2).to(torch.float64) * w).sum().item())
    tol = 1e-10 if dt in 'dD' else 1e-4
    assert abs(e_phys - e_spec) <= tol * e_phys, (e_phys, e_spec)
    back = fft.backward()
    rt = float(((back.tensor - u0).abs() ** 2).sum().sqrt().item() / (u0.abs() ** 2).sum().sqrt().item())
    assert rt <= tol, rt
    # a plane wave lands in exactly one bin with amplitude 1
    kx, ky, kz = 3, n // 2 - 1, 5
    x = torch.arange(n, device='cuda', dtype=torch.float64)
    ph = 2 * np.pi * (kx * x[:, None, None] + ky * x[None, :, None] + kz * x[None, None, :]) / n
    if dt in 'DF':
        u.tensor.copy_(torch.polar(torch.ones_like(ph), ph).to(u.tensor.dtype))
        amp = 1.0
    else:
        u.tensor.copy_(torch.cos(ph).to(u.tensor.dtype))
        amp = 0.5
    del ph
    uh = fft.forward(u)
    peak = complex(uh.tensor[kx, ky, kz].item())
    assert abs(peak - amp) <= (1e-12 if dt in 'dD' else 1e-5), peak
    uh.tensor[kx, ky, kz] = 0
    if dt in 'DF':
        rest = float(uh.tensor.abs().max().item())
    else:
        rest = float(uh.tensor.abs().max().item())
    assert rest <= (1e-12 if dt in 'dD' else 1e-5), rest
    fft.destroy()


def test_1024cubed_roundtrip_c128():
    """The headline configuration: 1024^3 complex128 forward -> backward, rel-err <= 1e-10."""
    import torch
    free, total = torch.cuda.mem_get_info()
    if free < 100 * 2 ** 30:
        pytest.skip('needs ~100 GiB of HBM')
    from mpi4py_fft_amd import PFFT, comm
    n = 1024
    fft = PFFT(comm.COMM_SELF, (n, n, n), dtype='D')
    u = fft.forward.input_array
    g = torch.Generator(device='cuda').manual_seed(1)
    ur = torch.view_as_real(u.tensor)
    for i in range(0, n, 64):       # fill in slabs to bound temporary memory
        ur[i:i + 64].copy_(torch.randn(ur[i:i + 64].shape, generator=g, device='cuda', dtype=torch.float64))
    u0 = u.tensor.clone()
    uh = fft.forward()
    e_phys = float((torch.view_as_real(u0) ** 2).sum().item()) / u0.numel()
    e_spec = float((torch.view_as_real(uh.tensor) ** 2).sum().item())
    assert abs(e_phys - e_spec) <= 1e-10 * e_phys
    back = fft.backward()
    num = float(((torch.view_as_real(back.tensor) - torch.view_as_real(u0)) ** 2).sum().sqrt().item())
    den = float((torch.view_as_real(u0) ** 2).sum().sqrt().item())
    assert num / den <= 1e-10, num / den
    fft.destroy()


@pytest.mark.parametrize('shape', [(128, 128, 128), (64, 128, 256), (256, 64, 128), (128, 256, 64), (32, 512, 1024)])
@pytest.mark.parametrize('dt', list('DdFf'))
def test_single_rank_3d_schedule_vs_oracle(shape, dt):
    """The fused single-GPU schedule (reordered passes + padded workspace, plan.cpp:plan_fused3)
    on non-cubic shapes: forward and backward values against the oracle, with the schedule forced
    on regardless of size."""
    from mpi4py_fft_amd import PFFT, newDistArray, comm, _lib
    import scipy.fft
    _lib.set_option('fused3_min_mib', 0)
    try:
        fft = PFFT(comm.COMM_SELF, shape, dtype=dt)
        assert 'padded-pitch workspace' in fft._fused_plans[0]._eng.plan_describe(fft._fused_plans[0]._plan)
        G = O.rng_array(shape, dt, 99)
        u = newDistArray(fft, False)
        u[...] = G
        uh = np.asarray(fft.forward(u)).copy()
        G64 = G.astype('D' if dt in 'DF' else 'd')
        ref = (scipy.fft.fftn(G64, workers=-1) if dt in 'DF' else scipy.fft.rfftn(G64, workers=-1)) / G.size
        tol = 2e-10 if dt in 'dD' else 2e-4
        assert uh.shape == ref.shape
        assert np.abs(uh - ref).max() <= tol * np.abs(ref).max()
        assert np.array_equal(np.asarray(u), G)
        vh = newDistArray(fft, True)
        vh[...] = ref.astype(uh.dtype)
        back = np.asarray(fft.backward(vh))
        assert np.linalg.norm(back - G) / np.linalg.norm(G) <= (1e-10 if dt in 'dD' else 1e-4)
        fft.destroy()
    finally:
        _lib.set_option('fused3_min_mib', 32)
