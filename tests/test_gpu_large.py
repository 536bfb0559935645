"""Size-independent properties at (near) BASELINE sizes, plus oracle comparisons at sizes the
oracle finishes in seconds."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import pfft_oracle as O
from tests import cases


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
    assert rt <= tol and rt <= cases.rounding_tol(dt, float(n) ** 3), rt
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


def _free_hbm():
    """Bytes of HBM free once what earlier tests left behind -- torch's cache, libgfft's per-stream workspaces -- is returned."""
    import gc
    import torch
    from mpi4py_fft_amd import _lib
    gc.collect()
    torch.cuda.empty_cache()
    _lib.lib().gfft_scratch_release()
    return torch.cuda.mem_get_info()[0]


def test_1024cubed_roundtrip_c128():
    """The headline configuration: 1024^3 complex128 forward -> backward, rel-err <= 1e-10."""
    import torch
    free = _free_hbm()
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
    assert 'fused pair' in fft._fused_plans[0]._eng.plan_describe(fft._fused_plans[0]._plan)      # the headline plan itself
    e_phys = float((torch.view_as_real(u0) ** 2).sum().item()) / u0.numel()
    e_spec = float((torch.view_as_real(uh.tensor) ** 2).sum().item())
    assert abs(e_phys - e_spec) <= 1e-10 * e_phys
    # six lines of the full-size forward against the DFT by definition: the (k0, k1) sums in float64 on the device,
    # the transform along axis 2 in long double (oracle/dft_oracle.c)
    from tests.test_gpu_c5 import _c_dft
    dft = _c_dft()
    for k0, k1 in [(3, 5), (n - 1, n - 1), (n // 2, 1), (n // 2 + 7, n // 4 + 3), (0, n // 2), (17, 0)]:
        want = dft(_line_partial(u0, (0, 0, 0), k0, k1, n, n), -1, n, 'D') / float(n) ** 3
        got = uh.tensor[k0, k1].cpu().numpy()
        assert np.abs(got - want).max() <= cases.tol_for('D', float(n) ** 3) * np.abs(want).max(), (k0, k1, np.abs(got - want).max() / np.abs(want).max())
    back = fft.backward()
    num = float(((torch.view_as_real(back.tensor) - torch.view_as_real(u0)) ** 2).sum().sqrt().item())
    den = float((torch.view_as_real(u0) ** 2).sum().sqrt().item())
    assert num / den <= 1e-10, num / den
    fft.destroy()


@pytest.mark.parametrize('n', [960, 896, 840])
def test_unequal_width_cubes_at_full_size_c128(n):
    """Round 5: n^3 complex128 at n = 960 (15 x 16 x 4), 896 (7 x 16 x 8), 840 (12 x 10 x 7) -- one pass per axis on stages of unequal
    width, the fused [axis 0 -> rows] pair at 960 / 896 -- at full size: Parseval, six lines against the DFT by definition (float64
    sums on the device, the last axis in long double: oracle/dft_oracle.c), round trip <= 1e-10."""
    import torch
    free = _free_hbm()
    if free < 80 * 2 ** 30:
        pytest.skip('needs ~80 GiB of HBM')
    from mpi4py_fft_amd import PFFT, comm
    fft = PFFT(comm.COMM_SELF, (n, n, n), dtype='D')
    desc = fft._fused_plans[0]._eng.plan_describe(fft._fused_plans[0]._plan)
    assert ('fused pair' in desc) == (n in (960, 896)), desc
    assert desc.count('n=%d' % n) >= 2 and 'n=48' not in desc and 'n=128 ' not in desc, desc       # one pass per axis
    u = fft.forward.input_array
    g = torch.Generator(device='cuda').manual_seed(n)
    ur = torch.view_as_real(u.tensor)
    for i in range(0, n, 64):
        ur[i:i + 64].copy_(torch.randn(ur[i:i + 64].shape, generator=g, device='cuda', dtype=torch.float64))
    u0 = u.tensor.clone()
    uh = fft.forward()
    e_phys = float((torch.view_as_real(u0) ** 2).sum().item()) / u0.numel()
    e_spec = float((torch.view_as_real(uh.tensor) ** 2).sum().item())
    assert abs(e_phys - e_spec) <= 1e-10 * e_phys
    from tests.test_gpu_c5 import _c_dft
    dft = _c_dft()
    for k0, k1 in [(3, 5), (n - 1, n - 1), (n // 2, 1), (n // 2 + 7, n // 4 + 3), (0, n // 2), (17, 0)]:
        want = dft(_line_partial(u0, (0, 0, 0), k0, k1, n, n), -1, n, 'D') / float(n) ** 3
        got = uh.tensor[k0, k1].cpu().numpy()
        assert np.abs(got - want).max() <= cases.tol_for('D', float(n) ** 3) * np.abs(want).max(), (k0, k1, np.abs(got - want).max() / np.abs(want).max())
    back = fft.backward()
    num = float(((torch.view_as_real(back.tensor) - torch.view_as_real(u0)) ** 2).sum().sqrt().item())
    den = float((torch.view_as_real(u0) ** 2).sum().sqrt().item())
    assert num / den <= 1e-10, num / den
    fft.destroy()


@pytest.mark.parametrize('shape,dt', [((960, 960, 960), 'd'), ((896, 896, 896), 'd'), ((1024, 1024, 2048), 'd'), ((960, 960, 960), 'F')])
def test_full_size_forward_lines_of_real_and_fp32_plans(shape, dt):
    """mpi4py-fft_amd/selftest.forward_gate (what bench.py admits plans on) on the reference's DEFAULT dtype (real,
    mpifft.py:202) at full size: six lines of the half spectrum against the DFT by definition in float64 on the device, for the
    round-5 shapes -- real rows of 960 / 896 entries on unequal-width stages, the 1025-wide half spectrum on its new pitch."""
    import torch
    from mpi4py_fft_amd import PFFT, comm, selftest
    free = _free_hbm()
    if free < 80 * 2 ** 30:
        pytest.skip('needs ~80 GiB of HBM')
    fft = PFFT(comm.COMM_SELF, shape, dtype=dt)
    u = fft.forward.input_array
    g = torch.Generator(device='cuda').manual_seed(7)
    ur = torch.view_as_real(u.tensor) if u.tensor.is_complex() else u.tensor
    for i in range(0, shape[0], 64):
        ur[i:i + 64].copy_(torch.randn(ur[i:i + 64].shape, generator=g, device='cuda', dtype=ur.dtype))
    u0 = u.tensor.clone()
    uh = fft.forward().tensor
    err = selftest.forward_gate(fft, comm.COMM_SELF, u0, uh)
    npts = float(np.prod(shape))
    assert err <= (2e-10 if dt in 'dD' else 2e-4) and err <= cases.rounding_tol(dt, npts), err
    back = fft.backward().tensor
    num = float(((back - u0).abs() ** 2).sum().sqrt().item())
    den = float((u0.abs() ** 2).sum().sqrt().item())
    assert num / den <= (1e-10 if dt in 'dD' else 1e-4) and num / den <= cases.rounding_tol(dt, npts), num / den
    fft.destroy()


# ((64, 128, 2048) / (32, 64, 4096): 1025- and 2049-wide half spectra and 2048-long complex rows, whose line-rounded
# workspace pitches used to be 129 x 2^k entries -- the pitch rule of plan_fused3)
@pytest.mark.parametrize('shape', [(128, 128, 128), (64, 128, 256), (256, 64, 128), (128, 256, 64), (32, 512, 1024), (64, 128, 2048),
                                   (32, 64, 4096)])
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
        desc = fft._fused_plans[0]._eng.plan_describe(fft._fused_plans[0]._plan)
        assert 'padded-pitch workspace' in desc
        # the workspace pitch: whole 128-byte lines, never a multiple of 2 KiB, never 129 x 2^k entries (channel aliasing)
        import re
        pitch = int(re.search(r'rows (\d+) entries apart', desc).group(1))
        isz = {'D': 16, 'd': 16, 'F': 8, 'f': 8}[dt]
        odd = pitch
        while odd % 2 == 0:
            odd //= 2
        assert pitch >= fft.forward.output_array.shape[2] and (pitch * isz) % 128 == 0 and (pitch * isz) % 2048 and odd != 129, desc
        G = O.rng_array(shape, dt, 99)
        u = newDistArray(fft, False)
        u[...] = G
        uh = np.asarray(fft.forward(u)).copy()
        G64 = G.astype('D' if dt in 'DF' else 'd')
        ref = (scipy.fft.fftn(G64, workers=-1) if dt in 'DF' else scipy.fft.rfftn(G64, workers=-1)) / G.size
        assert uh.shape == ref.shape
        cases.assert_close(uh, ref, dt, G.size, (shape, dt))                 # contract tolerance and rounding level
        assert np.array_equal(np.asarray(u), G)
        vh = newDistArray(fft, True)
        vh[...] = ref.astype(uh.dtype)
        back = np.asarray(fft.backward(vh))
        cases.assert_roundtrip(back, G, dt, G.size, (shape, dt))
        fft.destroy()
    finally:
        _lib.set_option('fused3_min_mib', 32)


# ---- the multi-rank BASELINE configurations at FULL size, as thread-ranks on one GPU ------------------------
# C3 (512^3 complex128 on 2 ranks, slab) and C4 (1024^3 complex128 on 8 ranks: pencil grid (4,2,1), and the slab
# grid (8,1,1)) fit one 288 GB device with every rank a thread and tests/fake_rccl as the wire.  What this catches
# before the first real multi-GPU run: 32-bit offsets, pitches and chunk plans at the real sizes of every stage
# (csrc/plan.cpp refuses a side beyond 2^31 elements: none of the stages here may hit that).  The reference runs its
# distributed tests the same way at 2 and 4 ranks, tests/runtests.sh:21-36, tests/test_mpifft.py:144-177.
def _line_partial(x, start, k0, k1, n0, n1):
    """sum over the LOCAL (i0, i1) of x[i0, i1, :] e^{-2 pi i (k0 I0 / n0 + k1 I1 / n1)}, I = global indices: this
    rank's share of the (k0, k1) line before its transform along axis 2 (float64 on the device)."""
    import torch
    l0, l1 = x.shape[0], x.shape[1]
    i0 = torch.arange(start[0], start[0] + l0, device='cuda', dtype=torch.float64)
    i1 = torch.arange(start[1], start[1] + l1, device='cuda', dtype=torch.float64)
    w0 = torch.polar(torch.ones_like(i0), -2 * np.pi * ((k0 * i0) % n0) / n0)
    w1 = torch.polar(torch.ones_like(i1), -2 * np.pi * ((k1 * i1) % n1) / n1)
    acc = torch.zeros(x.shape[2], device='cuda', dtype=torch.complex128)
    for a in range(0, l0, 32):
        acc += torch.einsum('a,b,abc->c', w0[a:a + 32], w1, x[a:a + 32])
    return acc.cpu().numpy()


@pytest.mark.parametrize('name,P,n,grid', [('C3', 2, 512, None), ('C4', 8, 1024, None), ('C4-slab', 8, 1024, [8, 1, 1])])
def test_multi_rank_baseline_configs_at_full_size_on_thread_ranks(name, P, n, grid):
    import torch
    from tests import cases, thread_comm
    from tests.test_gpu_c5 import _c_dft
    import os, subprocess
    from mpi4py_fft_amd import PFFT, newDistArray, _lib
    import gc
    gc.collect()
    torch.cuda.empty_cache()
    _lib.lib().gfft_scratch_release()           # (workspaces earlier tests left in libgfft's per-stream pool)
    free, _ = torch.cuda.mem_get_info()
    need = 40 if n == 512 else 200
    if free < need * 2 ** 30:
        pytest.skip('needs ~%d GiB of HBM, %d free' % (need, free >> 30))
    here = os.path.dirname(os.path.abspath(__file__))
    src, so = os.path.join(here, 'fake_rccl', 'fake_rccl.cpp'), os.path.join(here, 'fake_rccl', 'libfake_rccl.so')
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(['/opt/rocm/bin/hipcc', '-O2', '-std=c++17', '-fPIC', '-shared', '-x', 'hip', '--offload-arch=gfx950', src, '-o', so])
    _lib.check_wire(_lib.lib().gfft_rccl_load(so.encode()))
    shape = (n, n, n)
    kw = {} if grid is None else {'grid': grid}
    lines = [(3, 5), (n - 1, n - 1), (n // 2, 1), (n // 2 + 7, n // 4 + 3), (0, n // 2), (17, 0)]

    def body(comm):
        r = comm.Get_rank()
        staged = PFFT(comm, shape, dtype='D', wire='torch', exchange='direct', **{k: list(v) for k, v in kw.items()})
        pin, pout = staged.pencil
        geo = dict(dims=[c.Get_size() for c in staged.subcomm], in_shape=tuple(pin.subshape), in_start=tuple(pin.substart),
                   mid_shape=tuple(staged.transfer[0].subshapeB) if staged.transfer else None,
                   out_shape=tuple(pout.subshape), out_start=tuple(pout.substart))
        u = newDistArray(staged, False)
        g = torch.Generator(device='cuda').manual_seed(4321 + r)
        ur = torch.view_as_real(u.tensor)
        for i in range(0, ur.shape[0], 32):
            ur[i:i + 32].copy_(torch.randn(ur[i:i + 32].shape, generator=g, device='cuda', dtype=torch.float64))
        def fingerprint(t):        # bit-exact: wrap-around sum and xor-fold of the raw 64-bit words
            w = torch.view_as_real(t).view(torch.int64).reshape(-1)
            return int(w.sum().item()), int((w ^ (w >> 17)).sum().item())
        u0 = u.tensor.clone()
        e_phys = float((torch.view_as_real(u0) ** 2).sum().item())
        # this rank's share of six output lines (before their transform along axis 2)
        parts = [_line_partial(u0, pin.substart, k0, k1, n, n) for k0, k1 in lines]
        a = staged.forward(u).tensor
        e_spec = float((torch.view_as_real(a) ** 2).sum().item())
        fa = fingerprint(a)
        pieces = []                # ... and the pieces of those lines this rank holds
        for k0, k1 in lines:
            j1 = k1 - pout.substart[1]
            pieces.append((pout.substart[2], a[k0, j1].cpu().numpy()) if 0 <= j1 < pout.subshape[1] else None)
        sb = staged.backward().tensor
        num = float(((torch.view_as_real(sb) - torch.view_as_real(u0)) ** 2).sum().item())
        fb = fingerprint(sb)
        assert staged.pipeline is None
        staged.destroy()
        del sb, a, u0, staged, pin, pout
        import gc
        gc.collect()
        torch.cuda.empty_cache()
        comm.barrier()
        piped = PFFT(comm, shape, dtype='D', wire='native', exchange='direct', **{k: list(v) for k, v in kw.items()})
        assert piped.pipeline is not None
        info = piped.pipeline.describe()
        same_f = fingerprint(piped.forward(u).tensor) == fa
        same_b = fingerprint(piped.backward().tensor) == fb
        piped.destroy()
        return geo, e_phys, e_spec, num, same_f, same_b, parts, pieces, info

    try:
        res = cases.run_ranks(P, body)
    finally:
        _lib.lib().gfft_rccl_load(None)
    # geometry: the reference's own (tests/golden/geometry.npz, Appendix A of SURVEY.md), or the oracle's for the slab grid
    ref = O.OPFFT(P, shape, dtype='D', **kw)
    gold = cases.load('geometry')[name + '_P%d' % P] if grid is None else None
    for r, (geo, *_rest) in enumerate(res):
        assert geo['in_shape'] == tuple(ref.pencil_in[r].subshape) and geo['in_start'] == tuple(ref.pencil_in[r].substart)
        assert geo['out_shape'] == tuple(ref.pencil_out[r].subshape) and geo['out_start'] == tuple(ref.pencil_out[r].substart)
        if gold is not None:
            assert list(gold[r][8]) == geo['dims']
            assert tuple(gold[r][0]) == geo['in_shape'] and tuple(gold[r][1]) == geo['in_start']
            if P > 2:
                assert tuple(gold[r][4]) == geo['mid_shape']
            assert tuple(gold[r][6]) == geo['out_shape'] and tuple(gold[r][7]) == geo['out_start']
    N = float(n) ** 3
    e_phys = sum(x[1] for x in res)
    e_spec = sum(x[2] for x in res)
    assert abs(e_phys / N - e_spec) <= 1e-10 * e_phys / N, (e_phys / N, e_spec)                # Parseval (forward carries 1/N)
    rt = np.sqrt(sum(x[3] for x in res) / e_phys)
    assert rt <= 1e-10 and rt < 1e-14, rt                                                      # the north-star round trip
    assert all(x[4] for x in res) and all(x[5] for x in res), 'pipelined path differs from the staged one'
    assert all(any(e['chunks'] > 1 for e in x[8]) for x in res), res[0][8]
    # six lines of the forward transform against the DFT by definition in long double (oracle/dft_oracle.c)
    dft = _c_dft()
    for li, (k0, k1) in enumerate(lines):
        y = sum(x[6][li] for x in res)
        want = dft(y, -1, n, 'D') / N
        got = np.zeros(n, dtype='D')
        seen = np.zeros(n, dtype=bool)
        for x in res:
            if x[7][li] is not None:
                s2, piece = x[7][li]
                got[s2:s2 + piece.shape[0]] = piece
                seen[s2:s2 + piece.shape[0]] = True
        assert seen.all()
        assert np.abs(got - want).max() <= cases.tol_for('D', N) * np.abs(want).max(), (name, k0, k1, np.abs(got - want).max() / np.abs(want).max())


def test_c5_at_full_size_on_thread_ranks():
    """BASELINE config C5 whole: 2048^3 real fp32 -> r2c on 8 thread-ranks, grid (4,2,1), the Hermitian axis split 513 | 512
    (uneven exchange, pencil.py:5-9).  Staged wire (device copies between thread-ranks).  Geometry against the reference's
    (tests/golden/geometry.npz), Parseval over the half spectrum, round trip, four lines against the long-double DFT."""
    import torch
    from tests import cases
    from tests.test_gpu_c5 import _c_dft
    from mpi4py_fft_amd import PFFT, newDistArray, _lib
    import gc
    gc.collect()
    torch.cuda.empty_cache()
    _lib.lib().gfft_scratch_release()
    free, _ = torch.cuda.mem_get_info()
    if free < 260 * 2 ** 30:
        pytest.skip('needs ~260 GiB of HBM, %d free' % (free >> 30))
    n, P = 2048, 8
    shape = (n, n, n)
    lines = [(3, 5), (n - 1, n - 1), (n // 2 + 7, n // 4 + 3), (17, 0)]

    def body(comm):
        r = comm.Get_rank()
        fft = PFFT(comm, shape, dtype='f', wire='torch', exchange='direct')
        pin, pout = fft.pencil
        geo = dict(dims=[c.Get_size() for c in fft.subcomm], in_shape=tuple(pin.subshape), in_start=tuple(pin.substart),
                   mid_shape=tuple(fft.transfer[0].subshapeB), out_shape=tuple(pout.subshape), out_start=tuple(pout.substart))
        u = fft.forward.input_array
        g = torch.Generator(device='cuda').manual_seed(977 + r)
        for i in range(0, u.tensor.shape[0], 16):
            u.tensor[i:i + 16].copy_(torch.randn(u.tensor[i:i + 16].shape, generator=g, device='cuda', dtype=torch.float32))
        e_phys = sum(float((u.tensor[i:i + 64].double() ** 2).sum().item()) for i in range(0, u.tensor.shape[0], 64))
        parts = []
        for k0, k1 in lines:
            acc = torch.zeros(n, device='cuda', dtype=torch.complex128)
            i1 = torch.arange(pin.substart[1], pin.substart[1] + pin.subshape[1], device='cuda', dtype=torch.float64)
            w1 = torch.polar(torch.ones_like(i1), -2 * np.pi * ((k1 * i1) % n) / n)
            for a in range(0, pin.subshape[0], 16):
                i0 = torch.arange(pin.substart[0] + a, pin.substart[0] + a + 16, device='cuda', dtype=torch.float64)
                w0 = torch.polar(torch.ones_like(i0), -2 * np.pi * ((k0 * i0) % n) / n)
                acc += torch.einsum('a,b,abc->c', w0, w1, u.tensor[a:a + 16].to(torch.complex128))
            parts.append(acc.cpu().numpy())
        keep = u.tensor[:8].clone()                       # a slab of the input for the round-trip check (the planned input may be clobbered)
        uh = fft.forward().tensor
        wgt = torch.full((pout.subshape[2],), 2.0, device='cuda', dtype=torch.float64)
        k2 = torch.arange(pout.substart[2], pout.substart[2] + pout.subshape[2], device='cuda')
        wgt[(k2 == 0) | (k2 == n // 2)] = 1.0
        e_spec = sum(float(((torch.view_as_real(uh[i:i + 64]).double() ** 2).sum(-1) * wgt).sum().item()) for i in range(0, uh.shape[0], 64))
        pieces = []
        for k0, k1 in lines:
            j1 = k1 - pout.substart[1]
            pieces.append((pout.substart[2], uh[k0, j1].cpu().numpy()) if 0 <= j1 < pout.subshape[1] else None)
        back = fft.backward().tensor
        num = float(((back[:8].double() - keep.double()) ** 2).sum().item())
        den = float((keep.double() ** 2).sum().item())
        fft.destroy()
        return geo, e_phys, e_spec, num, den, parts, pieces

    res = cases.run_ranks(P, body)
    gold = cases.load('geometry')['C5_P8']
    for r, (geo, *_rest) in enumerate(res):
        assert list(gold[r][8]) == geo['dims']
        assert tuple(gold[r][0]) == geo['in_shape'] and tuple(gold[r][1]) == geo['in_start']
        assert tuple(gold[r][4]) == geo['mid_shape']
        assert tuple(gold[r][6]) == geo['out_shape'] and tuple(gold[r][7]) == geo['out_start']
    N = float(n) ** 3
    e_phys, e_spec = sum(x[1] for x in res), sum(x[2] for x in res)
    assert abs(e_phys / N - e_spec) <= 1e-5 * e_phys / N, (e_phys / N, e_spec)
    rt = np.sqrt(sum(x[3] for x in res) / sum(x[4] for x in res))
    assert rt <= 1e-4 and rt < 5e-6 and rt <= cases.rounding_tol('f', N), rt
    dft = _c_dft()
    for li, (k0, k1) in enumerate(lines):
        y = sum(x[5][li] for x in res)
        want = dft(y, -1, n, 'D')[:n // 2 + 1] / N
        got = np.zeros(n // 2 + 1, dtype='D')
        seen = np.zeros(n // 2 + 1, dtype=bool)
        for x in res:
            if x[6][li] is not None:
                s2, piece = x[6][li]
                got[s2:s2 + piece.shape[0]] = piece
                seen[s2:s2 + piece.shape[0]] = True
        assert seen.all()
        assert np.abs(got - want).max() <= cases.tol_for('f', N) * np.abs(want).max(), (k0, k1, np.abs(got - want).max() / np.abs(want).max())
