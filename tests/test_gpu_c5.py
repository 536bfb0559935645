"""BASELINE config C5 -- PFFT 3-D r2c fp32, pencil grid (4,2,1) on 8 ranks, padded Hermitian axis
and collapsed axes -- through the product path on one GPU (thread-ranks stand in for processes).

Small images of the 2048^3 case are compared value by value with the oracle; the per-rank pieces
of the real thing, (512,1024,2048) f32 -> (512,1024,1025) c64 -> (512,2048,513) -> (2048,512,513)
(SURVEY.md Appendix A), run at full size and are checked through size-independent properties plus
a spot check of single lines against the O(n^2) long-double DFT of oracle/dft_oracle.c.
Mirrors /root/reference/tests/test_mpifft.py:181-251 (padding / collapse loops) for dtype 'f'.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from tests import cases
from oracle import pfft_oracle as O

FP32_FWD_TOL = 2e-4     # forward vs oracle: max|d| <= tol * max|ref|   (cases.tol_for): the contract ...
FP32_RT_TOL = 1e-4      # round trip ||bwd(fwd(u)) - u|| / ||u||
FP32_LINE_GUARD = cases.rounding_tol('f', 2048)          # ... and what a 2048-point fp32 line may lose to rounding (5.2e-6)


@pytest.mark.parametrize('P', [1, 4, 8])
def test_c5_fp32_real_padding(P):
    """'f' + padding=[1.5]*3: padded r2c axis first (Hermitian truncation), padded c2c axes after."""
    cases.check_pfft_vs_oracle(P, (16, 16, 32), 'f', padding=[1.5, 1.5, 1.5])
    cases.check_pfft_vs_oracle(P, (32, 24, 20), 'f', padding=[1.5, 1.5, 1.5])
    if P == 8:
        cases.check_pfft_vs_oracle(P, (64, 64, 64), 'f', padding=[1.5, 1.5, 1.5], grid=[4, 2, 1])


@pytest.mark.parametrize('P', [1, 4, 8])
def test_c5_fp32_real_collapse(P):
    """'f' + collapse=True: trailing undistributed groups merge into one serial transform."""
    cases.check_pfft_vs_oracle(P, (16, 16, 32), 'f', collapse=True)
    cases.check_pfft_vs_oracle(P, (32, 24, 20), 'f', collapse=True, axes=((0,), (1,), (2,)))
    # slab grid: two axes collapse into one r2c plan
    cases.check_pfft_vs_oracle(P, (32, 24, 20), 'f', collapse=True, grid=[-1])
    cases.check_pfft_vs_oracle(P, (64, 32, 64), 'f', collapse=True, grid=[-1])


@pytest.mark.parametrize('P', [4, 8])
def test_c5_fp32_padding_and_collapse_together(P):
    # (padded groups must stay single-axis, libfft.py:389-390; on pencil grids nothing merges)
    cases.check_pfft_vs_oracle(P, (16, 16, 32), 'f', padding=[1.5, 1.5, 1.5], collapse=True)
    cases.check_pfft_vs_oracle(P, (20, 24, 16), 'f', padding=[1.5, 1.5, 1.5], collapse=True)


@pytest.mark.parametrize('shape', [(16, 16, 64), (32, 16, 128), (16, 32, 60)])
def test_c5_uneven_hermitian_split_at_grid_4x2(shape):
    """The 2048 -> 1025 -> 513 + 512 split in small: r2c half spectrum n/2+1 is odd, so the p=2
    exchange of the first redistribution is uneven (alltoallv counts), grid (4,2,1)."""
    from mpi4py_fft_amd import PFFT
    from tests import thread_comm
    nh = shape[2] // 2 + 1

    def geometry(comm):
        fft = PFFT(comm, shape, dtype='f', grid=[4, 2, 1])
        t = fft.transfer[0]
        out = (t.comm.Get_size(), t.subshapeB[2], fft.forward.output_array.shape, t.packedA, t.packedB)
        fft.destroy()
        return out
    geo = thread_comm.run(8, geometry)
    assert [g[0] for g in geo] == [2] * 8
    # the r2c kernel writes the uneven 513 | 512-style blocks of the exchange buffer itself whenever
    # the real length has a packed-real plan (gfft_plan_set_split on MODE_R2C_H): no pack kernel
    assert all(g[3] == (shape[2] in (64, 128)) for g in geo), [g[3] for g in geo]
    big, small = nh - nh // 2, nh // 2
    assert [g[1] for g in geo] == [big, small] * 4      # rank r sits in grid column r % 2
    assert geo[7][2] == (shape[0], shape[1] // 4, small)
    cases.check_pfft_vs_oracle(8, shape, 'f', grid=[4, 2, 1])
    cases.check_pfft_vs_oracle(8, shape, 'd', grid=[4, 2, 1])
    cases.check_pfft_vs_oracle(8, shape, 'f', grid=[4, 2, 1], padding=[1.5, 1.5, 1.5])


def test_c5_default_grid_is_4x2_and_matches_appendix_a():
    """Geometry of the real configuration (no arrays allocated): 2048^3 'f' on 8 ranks."""
    ref = O.OPFFT(8, (2048, 2048, 2048), dtype='f')
    assert [int(d) for d in ref.dims] == [4, 2, 1]
    pin, pout = ref.pencil_in[7], ref.pencil_out[7]
    assert tuple(pin.subshape) == (512, 1024, 2048)
    assert tuple(pout.subshape) == (2048, 512, 512) and tuple(pout.substart) == (0, 1536, 513)
    assert tuple(ref.pencil_out[6].subshape) == (2048, 512, 513)


# ---- the per-rank pieces of C5@8 at full size ------------------------------------------------

def _c_dft():
    import ctypes
    import os
    import subprocess
    here = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'oracle')
    so = os.path.join(here, 'libdft_oracle.so')
    if not os.path.exists(so):
        subprocess.check_call(['make', '-C', here])
    lib = ctypes.CDLL(so)
    i64p, dp = ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_double)
    lib.dft_oracle_xfftn.argtypes = [ctypes.c_int, i64p, dp, i64p, dp, ctypes.c_int,
                                     ctypes.POINTER(ctypes.c_int), ctypes.c_int]

    def run(line, kind, n_out, out_dtype):
        a = np.ascontiguousarray(line, dtype='D' if line.dtype.kind == 'c' else 'd')
        out = np.zeros((n_out,), dtype=out_dtype)
        si, so_ = (ctypes.c_int64 * 1)(a.shape[0]), (ctypes.c_int64 * 1)(n_out)
        ax = (ctypes.c_int * 1)(0)
        rc = lib.dft_oracle_xfftn(1, si, a.view('d').ctypes.data_as(dp), so_,
                                  out.view('d').ctypes.data_as(dp), 1, ax, kind)
        assert rc == 0
        return out
    return run


def _fill(t, seed):
    import torch
    g = torch.Generator(device='cuda').manual_seed(seed)
    r = torch.view_as_real(t) if t.is_complex() else t
    step = max(1, r.shape[0] // 8)
    for i in range(0, r.shape[0], step):       # in slabs, to bound temporary memory
        r[i:i + step].copy_(torch.randn(r[i:i + step].shape, generator=g, device='cuda', dtype=r.dtype))


def _energy(t):
    import torch
    r = torch.view_as_real(t) if t.is_complex() else t
    return float((r.to(torch.float64) ** 2).sum().item()) if r.numel() < (1 << 28) else \
        sum(float((r[i:i + 64].to(torch.float64) ** 2).sum().item()) for i in range(0, r.shape[0], 64))


def _need(gib):
    import torch
    free, _ = torch.cuda.mem_get_info()
    if free < gib * 2 ** 30:
        pytest.skip('needs ~%d GiB of HBM' % gib)


def test_c5_local_piece_r2c_rows_2048_full_size():
    """(512,1024,2048) f32 --r2c axis 2--> (512,1024,1025) c64: the first stage of every C5 rank."""
    import torch
    _need(24)
    from mpi4py_fft_amd.libfft import FFT
    shape = (512, 1024, 2048)
    f = FFT(shape, axes=(2,), dtype='f')
    u, uh = f.forward.input_array, f.forward.output_array
    assert uh.shape == (512, 1024, 1025) and uh.dtype == np.complex64
    _fill(u.tensor, 11)
    u0 = u.tensor.clone()
    f.forward()
    # Parseval per line, summed: sum x^2 = n * (|X0|^2 + 2 sum |Xk|^2 + |Xn/2|^2) with the 1/n forward
    w = torch.full((1025,), 2.0, device='cuda', dtype=torch.float64)
    w[0] = w[-1] = 1.0
    e_spec = sum(float(((torch.view_as_real(uh.tensor[i:i + 32]).to(torch.float64) ** 2).sum(-1) * w).sum().item())
                 for i in range(0, 512, 32))
    e_phys = _energy(u0) / 2048
    assert abs(e_phys - e_spec) <= 1e-5 * e_phys, (e_phys, e_spec)
    # spot check: single lines against the long-double DFT by definition
    dft = _c_dft()
    rng = np.random.default_rng(5)
    for _ in range(6):
        i, j = int(rng.integers(512)), int(rng.integers(1024))
        x = u0[i, j].cpu().numpy()
        got = uh.tensor[i, j].cpu().numpy()
        ref = dft(x, -2, 1025, 'D') / 2048
        assert np.abs(got - ref).max() <= min(FP32_FWD_TOL, FP32_LINE_GUARD) * np.abs(ref).max(), (i, j, np.abs(got - ref).max() / np.abs(ref).max())
    assert torch.equal(u.tensor, u0)           # the input is preserved
    f.backward()
    d = sum(float(((u.tensor[i:i + 64] - u0[i:i + 64]).to(torch.float64) ** 2).sum().item()) for i in range(0, 512, 64))
    rt = np.sqrt(d / _energy(u0))
    assert rt <= FP32_RT_TOL and rt < 2e-6, rt
    f.destroy()


@pytest.mark.parametrize('shape,axis', [((512, 2048, 513), 1), ((2048, 512, 513), 0), ((512, 2048, 512), 1)])
def test_c5_local_piece_c2c_strided_2048_full_size(shape, axis):
    """The second and third stages of a C5 rank: c64 length-2048 transforms along a strided axis
    of an array whose rows are 513 wide (not line aligned) or 512 wide (the other grid column)."""
    import torch
    _need(24)
    from mpi4py_fft_amd.libfft import FFT
    f = FFT(shape, axes=(axis,), dtype='F')
    u, uh = f.forward.input_array, f.forward.output_array
    _fill(u.tensor, 13 + axis)
    u0 = u.tensor.clone()
    f.forward()
    e_phys = _energy(u0) / 2048
    e_spec = _energy(uh.tensor)
    assert abs(e_phys - e_spec) <= 1e-5 * e_phys, (e_phys, e_spec)
    dft = _c_dft()
    rng = np.random.default_rng(6)
    for k in range(6):
        idx = [int(rng.integers(s)) for s in shape]
        if k == 0:
            idx = [s - 1 for s in shape]           # the last column of the odd-width rows
        sl = tuple(slice(None) if a == axis else idx[a] for a in range(3))
        x = u0[sl].cpu().numpy()
        got = uh.tensor[sl].cpu().numpy()
        ref = dft(x, -1, 2048, 'D') / 2048
        assert np.abs(got - ref).max() <= min(FP32_FWD_TOL, FP32_LINE_GUARD) * np.abs(ref).max(), (sl, np.abs(got - ref).max() / np.abs(ref).max())
    f.backward()
    d = sum(float((torch.view_as_real(u.tensor[i:i + 64] - u0[i:i + 64]).to(torch.float64) ** 2).sum().item())
            for i in range(0, shape[0], 64))
    rt = np.sqrt(d / _energy(u0))
    assert rt <= FP32_RT_TOL and rt < 2e-6, rt
    f.destroy()


def test_c5_uneven_exchange_buffers_at_full_width():
    """Pack / unpack of the first C5 redistribution at full line width: (64,1024,1025) c64 cut into
    513 + 512 along the Hermitian axis, against numpy slicing."""
    import torch
    from mpi4py_fft_amd import _lib
    shape = (16, 256, 1025)
    eng = _lib.engine()
    a = torch.randn(shape + (2,), device='cuda', dtype=torch.float32)
    a = torch.view_as_complex(a)
    packed = torch.empty_like(a)
    eng.pack(a, packed, shape, 2, 2, 8)
    h = a.cpu().numpy()
    want = np.concatenate([h[:, :, :513].ravel(), h[:, :, 513:].ravel()])
    assert np.array_equal(packed.cpu().numpy().ravel(), want)
    back = torch.empty_like(a)
    eng.unpack(packed, back, shape, 2, 2, 8)
    assert torch.equal(back, a)
