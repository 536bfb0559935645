"""GPU parity of the serial transforms, called through the C ABI (fftw planners -> libgfft.so),
against the oracle (numpy pocketfft, pinned to the reference by test_oracle_golden.py).
Tolerances: fp64 forward max|d| <= 2e-10 max|ref| and round trip <= 1e-10;
in practice ~1e-15.  fp32: 2e-4 / 1e-4."""
import itertools

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import pfft_oracle as O


def _tol(dt):
    return 2e-10 if dt in 'dD' else 2e-4


def _check(shape, axes, dt, seed=0):
    from mpi4py_fft_amd import FFT, asdevice
    fft = FFT(shape, axes, dtype=dt)
    ref = O.OFFT(shape, axes, dt)
    A = O.rng_array(shape, dt, seed)
    B = np.asarray(fft.forward(asdevice(A)))
    Bref = ref.forward(A)
    assert B.shape == Bref.shape and B.dtype == Bref.dtype
    err = np.abs(B - Bref).max() / max(np.abs(Bref).max(), 1e-30)
    assert err <= _tol(dt), (shape, axes, dt, 'fwd', err)
    A2 = np.asarray(fft.backward(asdevice(Bref)))
    rt = np.linalg.norm(A2 - A) / np.linalg.norm(A)
    assert A2.shape == A.shape and A2.dtype == A.dtype
    assert rt <= (1e-10 if dt in 'dD' else 1e-4), (shape, axes, dt, 'round trip', rt)
    # tighter than the contract: both precisions at rounding level (tests/cases.py rounding_tol)
    if dt in 'dD':
        assert err < 1e-13 and rt < 1e-13, (shape, axes, err, rt)
    from tests import cases
    npts = int(np.prod([shape[a] for a in (axes if np.ndim(axes) else [axes])])) if axes is not None else int(np.prod(shape))
    cases._note('fwd', dt, npts, err)
    cases._note('rt', dt, npts, rt)
    guard = cases.rounding_tol(dt, npts)
    assert err <= guard and rt <= guard, (shape, axes, dt, 'ROUNDING-LEVEL guard', err, rt, guard)
    fft.destroy()


def test_docstring_kats(golden):
    from mpi4py_fft_amd import fftw
    k = golden['libfft']
    A = fftw.aligned(4, dtype='D')
    plan = fftw.fftn(A, flags=(fftw.FFTW_ESTIMATE,))
    A[:] = k['kat/fftn_in']
    B = plan()
    assert np.allclose(B, k['kat/fftn_out'], atol=1e-14)
    assert plan.input_array is A and plan.output_array is B
    A = fftw.aligned(4, dtype='d')
    plan = fftw.rfftn(A)
    A[:] = k['kat/rfftn_in']
    assert np.allclose(plan(), k['kat/rfftn_out'], atol=1e-14)
    A = fftw.aligned(4, dtype='D')
    plan = fftw.irfftn(A)
    A[:] = k['kat/irfftn_in']
    assert np.allclose(plan(), k['kat/irfftn_out6'], atol=1e-13)
    plan = fftw.irfftn(A, s=(7,))
    A[:] = k['kat/irfftn_in']
    assert np.allclose(plan(), k['kat/irfftn_out7'], atol=1e-7)
    plan = fftw.fftn(A)
    A[:] = k['kat/fftn_in']
    assert np.allclose(plan(normalize=True), k['kat/fftn_out'] / 4, atol=1e-14)


@pytest.mark.parametrize('dt', list('dDfF'))
def test_libfft_loop(dt):
    """tests/test_libfft.py:24-64: dims 1-3, sizes (7,8,9), contiguous axis subsets."""
    sizes = (7, 8, 9)
    for dim in (1, 2, 3):
        for shape in itertools.product(*([sizes] * dim)):
            allaxes = tuple(reversed(range(dim)))
            for i in range(dim):
                for j in range(i + 1, dim):
                    for axes in (None, allaxes[i:j]):
                        _check(shape, axes, dt)


@pytest.mark.parametrize('dt', ['D', 'F'])
@pytest.mark.parametrize('n', [16, 32, 64, 128, 256, 512, 1024, 2048, 4096])
def test_pow2_rows_and_cols(n, dt):
    _check((5, n), (1,), dt)          # contiguous axis: ROWS kernels, ragged tile
    _check((n, 20), (0,), dt)         # strided axis: COLS kernels, ragged tile (20 % 8, 20 % 16)
    _check((3, n, 9), (1,), dt)       # strided axis in the middle, odd inner


@pytest.mark.parametrize('dt', ['d', 'f'])
@pytest.mark.parametrize('n', [16, 64, 256, 1024, 2048])
def test_pow2_real(n, dt):
    _check((6, n), (1,), dt)
    _check((n, 12), (0,), dt)
    _check((4, n, 6), (1, 0), dt)


@pytest.mark.parametrize('shape,axes', [
    ((2, 3), None), ((1, 5), (1,)), ((13,), None), ((12, 13), (0, 1)), ((12, 13), (1, 0)),
    ((100, 3), (0,)), ((3, 100), (1,)), ((6, 35, 4), (1,)), ((243, 2), (0,)), ((2, 625), (1,)),
    ((4, 1001), (1,)), ((1000, 5), (0,)), ((17, 19, 23), None), ((30, 30, 30), (2, 0, 1)),
    ((3000, 2), (0,)), ((2, 4095), (1,)), ((127, 3), (0,)), ((3, 509), (1,)),
])
@pytest.mark.parametrize('dt', ['D', 'd'])
def test_generic_sizes(shape, axes, dt):
    _check(shape, axes, dt)


@pytest.mark.parametrize('n', [8192, 16384, 5000, 10000, 1 << 18])
def test_four_step_1d(n):
    _check((3, n), (1,), 'D')
    _check((n, 3), (0,), 'D')


def test_four_step_2pow20_batch():
    """BASELINE config C2 (batched 1-D 2^20, complex128), at a batch the oracle finishes fast."""
    _check((4, 1 << 20), (1,), 'D')
    _check((2, 1 << 20), (1,), 'F')


def test_in_place_and_implicit_arrays():
    from mpi4py_fft_amd import fftw, asdevice
    A = O.rng_array((64, 48), 'D', 3)
    a = asdevice(A)
    plan = fftw.fftn(a, axes=(0, 1), output_array=a)       # in place
    plan()
    assert np.abs(np.asarray(a) - np.fft.fftn(A)).max() < 1e-12 * np.abs(A).max() * A.size ** 0.5
    b, c = asdevice(A), fftw.aligned(A.shape, dtype='D')
    plan2 = fftw.fftn(fftw.aligned(A.shape, dtype='D'), axes=(1,))
    out = plan2(b, c)                                      # implicit: runs directly on b -> c
    assert out is c
    assert np.allclose(np.asarray(c), np.fft.fft(A, axis=1), atol=1e-11)
    assert np.array_equal(np.asarray(b), A)                # out-of-place c2c preserves its input


def test_plan_failure_raises():
    from mpi4py_fft_amd import fftw
    with pytest.raises(RuntimeError):
        fftw.fftn(fftw.aligned((1, (1 << 26) + 1), dtype='D'), axes=(1,))   # 5 x 53 x 157 x 1613: Bluestein on 2^28 points, beyond its table limit


@pytest.mark.parametrize('n', [1 << 24, 1 << 25, 3 << 23, 5 << 23])
def test_lengths_of_2pow24_and_beyond(n):
    """FFTW plans any length (/root/reference/mpi4py_fft/fftw/fftw_planxfftn.c:52-75).  2^24 = 4096 x 4096 is a plain four-step
    transform; beyond it one more strided pass in front of a four-step transform of the cofactor (plan.cpp plan_long),
    three passes in all: against the oracle, both directions, as rows and along a strided axis."""
    from mpi4py_fft_amd import fftw, _lib
    a = fftw.aligned((1, n), dtype='D')
    plan = fftw.fftn(a, axes=(1,))
    desc = _lib.engine().plan_describe(plan._plan)
    assert desc.count('\n  n=') + 2 * desc.count('fused pair') == (2 if n == 1 << 24 else 3), desc          # HBM round trips
    assert ('FS2' in desc) == (n > 1 << 24), desc
    plan.destroy()
    _check((1, n), (1,), 'D')
    if n <= 1 << 25:
        _check((n, 2), (0,), 'D')


def test_long_lengths_real_single_precision_and_prime():
    _check((2, 1 << 25), (1,), 'F')
    _check((1, 1 << 25), (1,), 'd')             # real: as a complex line of the same length
    _check((1, 3 << 23), (1,), 'f')
    _check((1, 16777259), (1,), 'D')            # a prime beyond 2^24: Bluestein on 2^26 points, themselves a three-pass transform


def test_length_2pow28_against_the_dft_by_definition():
    """2^28 complex128 (4 GiB a side), too long for the CPU oracle inside a test: six output entries against the DFT sum,
    accumulated on the device in float64 with the phase reduced exactly (j k mod n in integers), and the round trip."""
    import torch
    from mpi4py_fft_amd import fftw, DeviceArray
    n = 1 << 28
    a = fftw.aligned((n,), dtype='D')
    out = fftw.aligned((n,), dtype='D')
    g = torch.Generator(device='cuda').manual_seed(7)
    torch.view_as_real(a.tensor).copy_(torch.randn((n, 2), dtype=torch.float64, device='cuda', generator=g))
    keep = a.tensor.clone()
    fwd = fftw.fftn(a, axes=(0,), output_array=out)
    bwd = fftw.ifftn(out, axes=(0,), output_array=fftw.aligned((n,), dtype='D'))
    fwd()
    j = torch.arange(n, dtype=torch.int64, device='cuda')
    scale = float(keep.abs().max().item()) * n ** 0.5
    for k in (0, 1, 12345, n // 2 + 3, n - 1, 177777777):
        ph = ((j * k) % n).to(torch.float64) * (-2.0 * np.pi / n)
        ref = torch.sum(keep * torch.complex(torch.cos(ph), torch.sin(ph)))
        got = out.tensor[k]
        assert abs(complex(got.item()) - complex(ref.item())) <= 1e-10 * scale, (k, got, ref)
        del ph
    back = bwd(out, normalize=True)
    assert float((back.tensor - keep).abs().max().item()) <= 1e-12 * float(keep.abs().max().item())
    fwd.destroy(); bwd.destroy()


@pytest.mark.parametrize('dt', ['D', 'd', 'F'])
def test_padding_truncation(dt):
    """tests/test_libfft.py:67-98: fwd . bwd . fwd idempotence + values vs the oracle."""
    from mpi4py_fft_amd import FFT, asdevice
    for padding in (1.5, 2.0):
        for shape in ((12, 9), (9, 12), (13, 8, 12), (18, 7)):
            for axis in range(len(shape)):
                shp = list(shape)
                shp[axis] = int(shp[axis] * padding)
                fft = FFT(shp, axis, dtype=dt, padding=padding)
                ref = O.OFFT(shp, axis, dt, padding=padding)
                A = O.rng_array(shp, dt, 11)
                B = np.asarray(fft.forward(asdevice(A))).copy()
                Bref = ref.forward(A)
                tol = _tol(dt)
                assert B.shape == Bref.shape
                assert np.abs(B - Bref).max() <= tol * np.abs(Bref).max(), (shp, axis, padding)
                A1 = np.asarray(fft.backward(asdevice(B))).copy()
                assert np.abs(A1 - ref.backward(Bref)).max() <= 10 * tol * np.abs(A).max()
                B2 = np.asarray(fft.forward(asdevice(A1)))
                assert np.abs(B2 - B).max() <= 10 * tol * np.abs(B).max()
                fft.destroy()


def test_pack_unpack_blocks():
    from mpi4py_fft_amd import _lib, asdevice, zeros
    eng = _lib.engine()
    rng = np.random.default_rng(0)
    for shape, axis, p, dt in (((7, 8, 9), 1, 3, 'd'), ((7, 8, 9), 2, 4, 'D'), ((5, 13), 0, 2, 'f'),
                               ((4, 9, 6), 0, 4, 'D'), ((3, 1025, 4), 1, 2, 'F'), ((16, 16, 16), 2, 8, 'D')):
        a = rng.standard_normal(shape).astype(dt)
        d = asdevice(a)
        packed = zeros(shape, dt)
        eng.pack(d.tensor, packed.tensor, shape, axis, p, a.itemsize)
        ref = np.concatenate([np.ascontiguousarray(
            np.take(a, range(s, s + n), axis=axis)).reshape(-1)
            for n, s in (O.blockdist(shape[axis], p, i) for i in range(p))])
        assert np.array_equal(np.asarray(packed).reshape(-1), ref), (shape, axis, p)
        back = zeros(shape, dt)
        eng.unpack(packed.tensor, back.tensor, shape, axis, p, a.itemsize)
        assert np.array_equal(np.asarray(back), a)


def test_callers_input_is_read_in_place_and_preserved():
    """PFFT.forward(u) / backward(uh) read a caller's device array directly (no copy-in) and never
    write it -- including multi-axis c2r, which FFTW would clobber."""
    from mpi4py_fft_amd import PFFT, newDistArray, comm, asdevice, FFT
    for shape, dt in (((16, 12, 10), 'd'), ((64, 64, 64), 'D'), ((12, 13), 'd'), ((256, 256, 256), 'd')):
        fft = PFFT(comm.COMM_SELF, shape, dtype=dt)
        G = O.rng_array(shape, dt, 3)
        u = newDistArray(fft, False)
        u[...] = G
        uh = fft.forward(u)
        assert np.array_equal(np.asarray(u), G)                      # input untouched
        ref = O.OPFFT(1, shape, dtype=dt).forward([G])[0]
        assert np.abs(np.asarray(uh) - ref).max() <= 1e-13 * np.abs(ref).max()
        vh = newDistArray(fft, True)
        vh[...] = ref
        back = fft.backward(vh)
        assert np.array_equal(np.asarray(vh), ref)                   # spectral input untouched
        assert np.abs(np.asarray(back) - G).max() <= 1e-12
        fft.destroy()
    # serial multi-axis c2r through libfft.FFT
    f = FFT((12, 10, 8), (0, 1, 2), dtype='d')
    A = O.rng_array((12, 10, 8), 'd', 1)
    B = np.asarray(f.forward(asdevice(A))).copy()
    b = asdevice(B)
    A2 = np.asarray(f.backward(b))
    assert np.array_equal(np.asarray(b), B) and np.abs(A2 - A).max() < 1e-12


MIX3 = [48, 72, 96, 108, 144, 192, 216, 288, 384, 432, 576, 768, 864, 1152, 1296, 1536, 1728, 2304, 2592, 3072, 3456]


@pytest.mark.parametrize('n', MIX3)
def test_mix3_rows_and_cols(n):
    """Lengths 3^b * 2^k on the register-resident kernels (R = 12): contiguous and strided axes,
    ragged tiles, both precisions, real and complex."""
    _check((3, n), (1,), 'D')
    _check((n, 20), (0,), 'D')
    _check((2, n, 5), (1,), 'F')
    _check((3, n), (1,), 'd')
    _check((n, 6), (0,), 'f')


@pytest.mark.parametrize('shape,dt', [((96, 48, 192), 'D'), ((192, 96, 144), 'd'), ((72, 64, 108), 'F'),
                                      ((384, 384, 384), 'D'), ((768, 96, 64), 'f')])
def test_mix3_3d(shape, dt):
    _check(shape, None, dt)


def test_four_step_mixed_lengths():
    _check((2, 3 * 4096), (1,), 'D')      # 12288 = 96 x 128
    _check((3 * 2048, 2), (0,), 'D')
    _check((2, 9 * 4096), (1,), 'F')


@pytest.mark.parametrize('n', [67, 127, 1021, 2042, 2053, 4099, 5003, 3 * 1021])
def test_bluestein_lengths(n):
    """Lengths with a large prime factor run through Bluestein (chirp -> FFT_M -> xB -> IFFT_M ->
    chirp) on the fast kernels; values against the oracle, all kinds."""
    _check((3, n), (1,), 'D')
    _check((n, 4), (0,), 'D')
    _check((2, n), (1,), 'd')
    _check((n, 3), (0,), 'd')
    _check((2, n), (1,), 'F')


@pytest.mark.parametrize('n', [8192, 6000, 12288, 10000])
def test_long_real_transforms(n):
    """r2c / c2r beyond the single-pass limit: complex embedding around the four-step engine."""
    _check((3, n), (1,), 'd')
    _check((n, 2), (0,), 'd')
    _check((2, n), (1,), 'f')


def test_plan_description_mentions_engines():
    from mpi4py_fft_amd import fftw
    a = fftw.aligned((4, 2053), dtype='D')
    p = fftw.fftn(a, axes=(1,))
    d = p._eng.plan_describe(p._plan)
    assert 'embed' in d and 'multiply by B' in d and 'extract' in d
    p2 = fftw.fftn(fftw.aligned((4, 16384), dtype='D'), axes=(1,))
    assert 'four-step' in p2._eng.plan_describe(p2._plan)


MIX5 = [20, 40, 80, 100, 160, 200, 320, 400, 500, 640, 800, 1000, 1280, 1600, 2000, 2500, 2560, 3200, 4000]


@pytest.mark.parametrize('n', MIX5)
def test_mix5_rows_and_cols(n):
    """Lengths 5^c * 2^k on the register-resident kernels (R = 20)."""
    _check((3, n), (1,), 'D')
    _check((n, 20), (0,), 'D')
    _check((2, n, 5), (1,), 'F')
    _check((3, n), (1,), 'd')
    _check((n, 6), (0,), 'f')


@pytest.mark.parametrize('n', [240, 480, 960, 1920, 3840, 1440, 2880, 720, 3600, 900])
def test_two_pass_235_lengths(n):
    """2^a 3^b 5^c lengths outside the single-pass tables: split into two register-kernel passes."""
    _check((3, n), (1,), 'D')
    _check((n, 10), (0,), 'D')
    _check((2, n), (1,), 'd')       # real: generic / embedding
    _check((2, n, 3), (1,), 'F')


MIXV_ALL = [98, 112, 120, 140, 150, 168, 180, 210, 224, 240, 250, 252, 280, 300, 336, 350, 360, 392, 420, 448, 450, 480, 490, 540, 560, 588,
            600, 672, 700, 720, 750, 784, 840, 896, 900, 960, 980, 1008, 1050, 1120, 1200, 1260, 1344, 1400, 1440, 1500, 1680, 1792, 1800,
            1920, 2100, 2160, 2240, 2250, 2400, 2700, 2800, 2880, 3000, 3584, 3600, 3840]      # (tools/gen_mixv_tables.py prints this list)


@pytest.mark.parametrize('dt', ['D', 'F'])
def test_every_generated_unequal_width_plan(dt):
    """All 62 lengths of csrc/fft_mixv_*.hip (generated: tools/gen_mixv_tables.py): one pass, rows and ragged strided tiles and
    real lines of twice the length, against the oracle."""
    from mpi4py_fft_amd import FFT, _lib
    for n in MIXV_ALL:
        for shape, axes, d in (((3, n), (1,), dt), ((n, 21), (0,), dt), ((5, 2 * n), (1,), dt.lower())):
            fft = FFT(shape, axes, dtype=d)
            desc = _lib.engine().plan_describe(fft.fwd._plan)
            assert desc.count('kernel=regs') == 1 and '1 passes' in desc, (n, desc)
            fft.destroy()
            _check(shape, axes, d, seed=n)


@pytest.mark.parametrize('n', [240, 480, 960, 1920, 3840, 720, 1440, 2880, 1200, 2400, 112, 224, 448, 896, 1792, 3584, 336, 600, 840, 1008, 1680, 3000, 3600])
@pytest.mark.parametrize('dt', ['D', 'F'])
def test_one_pass_kernels_for_3x5x2k_lengths(n, dt):
    """Round 5: lengths 3 x 5 x 2^k (and 9 x 5 x 2^k, 3 x 25 x 2^k) and 7 x 2^k as ONE register-kernel pass whose stages keep different
    numbers of values per thread (csrc/fft_mixv_*.hip, Geo / StageV): rows, strided columns (whole and ragged tiles),
    middle axis; values against the oracle; the planner says one pass; and the same lines through the two-pass plans of
    rounds 1-4 (option mixv = 0) agree to rounding.  The reference's tests exercise such sizes: tests/test_libfft.py:26-27."""
    from mpi4py_fft_amd import FFT, asdevice, _lib
    for shape, axes in (((3, n), (1,)), ((n, 10), (0,)), ((n, 37), (0,)), ((2, n, 19), (1,)), ((70, n), (1,))):
        fft = FFT(shape, axes, dtype=dt)
        desc = _lib.engine().plan_describe(fft.fwd._plan)
        assert desc.count('kernel=regs') == 1 and '1 passes' in desc, desc
        fft.destroy()
        _check(shape, axes, dt, seed=n)
    # real lines of 2 n entries: packed-real rows on the same row plan (Hermitian pass in the geometry of its side)
    rdt = dt.lower()
    for shape in ((3, 2 * n), (37, 2 * n)):
        fft = FFT(shape, (1,), dtype=rdt)
        desc = _lib.engine().plan_describe(fft.fwd._plan) + _lib.engine().plan_describe(fft.bck._plan)
        assert desc.count('kernel=regs') == 2 and desc.count('1 passes') == 2, desc
        fft.destroy()
        _check(shape, (1,), rdt, seed=n + 1)
    A = O.rng_array((5, n), dt, 3)
    got = {}
    for opt in (1, 0):
        _lib.set_option('mixv', opt)
        try:
            fft = FFT((5, n), (1,), dtype=dt)
            got[opt] = np.asarray(fft.forward(asdevice(A))).copy()
            fft.destroy()
        finally:
            _lib.set_option('mixv', 1)
    assert np.abs(got[0] - got[1]).max() <= (1e-13 if dt == 'D' else 1e-5) * np.abs(got[0]).max()


@pytest.mark.parametrize('shape,dt', [((240, 480, 240), 'D'), ((480, 240, 720), 'F'), ((240, 240, 960), 'd'), ((960, 240, 64), 'D'),
                                      ((240, 720, 480), 'f'), ((128, 240, 1920), 'd'), ((224, 448, 112), 'D'), ((112, 224, 896), 'd')])
def test_3x5x2k_lengths_in_the_one_rank_3d_schedule(shape, dt):
    """... and inside the single-GPU 3-D schedule (plan.cpp plan_fused3: pitched workspace, reordered passes); a real
    transform takes them on its two complex axes."""
    from mpi4py_fft_amd import _lib
    _lib.set_option('fused3_min_mib', 0)
    try:
        _check(shape, None, dt)
    finally:
        _lib.set_option('fused3_min_mib', 32)


@pytest.mark.parametrize('shape,dt', [((100, 80, 160), 'D'), ((200, 40, 100), 'd'), ((1000, 20, 40), 'F'),
                                      ((320, 200, 400), 'D')])
def test_mix5_3d(shape, dt):
    _check(shape, None, dt)


@pytest.mark.parametrize('dt', ['D', 'd', 'F', 'f'])
def test_fused_truncation_and_padding_on_unequal_width_lengths(dt):
    """Round 5: the 3/2-rule applied to 5 x 2^k / 7 x 2^k grids lands on 3 x 5 x 2^k / 3 x 7 x 2^k lengths (160 -> 240, 640 -> 960,
    448 -> 672); their kernels (csrc/fft_mixv_*.hip) carry the truncating store / zero-padding load where the length
    divides by 3 -- values against the oracle, even and odd kept lengths, rows / strided / real lines."""
    from mpi4py_fft_amd import FFT, asdevice
    # (shapes are the TRANSFORMED, i.e. padded, ones: 240 keeps 160, 960 keeps 640, 672 keeps 448 ...)
    for shp, axis, padding in (((240, 6), 0, 1.5), ((5, 960), 1, 1.5), ((4, 480, 3), 1, 1.5), ((672, 5), 0, 1.5), ((3, 1920), 1, 1.5),
                               ((2, 336), 1, 1.5), ((7, 240), 1, 240 / 161.), ((840, 20), 0, 1.5), ((2, 3600), 1, 1.5)):
        fft = FFT(shp, axis, dtype=dt, padding=padding)
        # (a REAL transform along a strided axis has no kernel on these lengths -- LDS kernel + separate truncation: still checked)
        assert fft._fused_trunc or (dt in 'df' and axis != len(shp) - 1), (shp, axis, padding)
        ref = O.OFFT(shp, axis, dt, padding=padding)
        A = O.rng_array(shp, dt, 23)
        B = np.asarray(fft.forward(asdevice(A))).copy()
        Bref = ref.forward(A)
        tol = _tol(dt)
        assert B.shape == Bref.shape and B.dtype == Bref.dtype
        assert np.abs(B - Bref).max() <= tol * np.abs(Bref).max(), (shp, axis, padding, dt)
        A1 = np.asarray(fft.backward(asdevice(Bref))).copy()
        assert np.abs(A1 - ref.backward(Bref)).max() <= 10 * tol * np.abs(A).max(), (shp, axis, padding, dt)
        fft.destroy()


@pytest.mark.parametrize('dt', ['D', 'd', 'F', 'f'])
def test_fused_truncation_and_padding(dt):
    """3/2-rule truncation / zero padding fused into the transform (gfft_plan_set_truncation) on
    register-kernel lengths: values vs the oracle, even and odd truncated lengths, contiguous and
    strided axes, and the fwd.bwd.fwd idempotence of tests/test_libfft.py:67-98."""
    from mpi4py_fft_amd import FFT, asdevice
    cases_ = [((48, 10), 0, 1.5), ((10, 48), 1, 1.5), ((6, 96, 5), 1, 1.5), ((64, 7), 0, 2.0),
              ((5, 192), 1, 1.5), ((48, 6), 0, 48 / 31.), ((4, 80), 1, 80 / 63.), ((1536, 3), 0, 1.5),
              ((3, 1536), 1, 1.5), ((100, 4), 0, 2.0), ((3, 768, 4), 1, 1.5), ((1024, 2), 0, 2.0)]
    for shp, axis, padding in cases_:
        fft = FFT(shp, axis, dtype=dt, padding=padding)
        # (5^c 2^k lengths carry the fused adapters only in a `make VARIANTS=1` library; otherwise the
        # separate truncate / pad kernels serve them -- same values either way)
        assert fft._fused_trunc or shp[axis] % 5 == 0, (shp, axis, padding)
        ref = O.OFFT(shp, axis, dt, padding=padding)
        A = O.rng_array(shp, dt, 21)
        B = np.asarray(fft.forward(asdevice(A))).copy()
        Bref = ref.forward(A)
        tol = _tol(dt)
        assert B.shape == Bref.shape and B.dtype == Bref.dtype
        assert np.abs(B - Bref).max() <= tol * np.abs(Bref).max(), (shp, axis, padding, dt)
        A1 = np.asarray(fft.backward(asdevice(Bref))).copy()
        assert np.abs(A1 - ref.backward(Bref)).max() <= 10 * tol * np.abs(A).max(), (shp, axis, padding, dt)
        B2 = np.asarray(fft.forward(asdevice(A1)))
        assert np.abs(B2 - B).max() <= 10 * tol * np.abs(B).max()
        fft.destroy()


def test_padded_pfft_uses_fused_truncation():
    from mpi4py_fft_amd import PFFT, newDistArray, comm
    shape = (32, 64, 128)
    for dt in 'dD':
        fft = PFFT(comm.COMM_SELF, shape, dtype=dt, padding=[1.5, 1.5, 1.5])
        assert all(x._fused_trunc for x in fft.xfftn)
        ref = O.OPFFT(1, shape, dtype=dt, padding=[1.5, 1.5, 1.5])
        G = O.rng_array(ref.input_shape, dt, 5)
        u = newDistArray(fft, False)
        u[...] = G
        uh = np.asarray(fft.forward(u)).copy()
        want = ref.forward([G])[0]
        assert np.abs(uh - want).max() <= 2e-10 * np.abs(want).max()
        back = np.asarray(fft.backward())
        assert np.abs(back - ref.backward([want])[0]).max() <= 1e-10 * max(1, np.abs(G).max())
        fft.destroy()


def _packed(a, axis, p):
    n = a.shape[axis]
    return np.concatenate([np.ascontiguousarray(np.take(a, range(s, s + m), axis=axis)).reshape(-1)
                           for m, s in (O.blockdist(n, p, i) for i in range(p))])


@pytest.mark.parametrize('dt', ['D', 'F'])
@pytest.mark.parametrize('shape,axis', [((6, 16, 5), 1), ((3, 5, 32), 2), ((64, 7, 3), 0), ((5, 256, 18), 1),
                                        ((4, 3, 1024), 2), ((1024, 3, 17), 0), ((2, 2048, 16), 1),
                                        ((3, 2, 4096), 2), ((3, 768, 8), 1), ((2, 5, 1000), 2),
                                        ((96, 4, 33), 0), ((2, 640, 9), 1), ((2, 512, 513), 1)])
def test_split_layout_plans(shape, axis, dt):
    """gfft_plan_set_split: a plan writes what gfft_pack would make of its natural output, and
    reads what gfft_unpack would consume (bit-identical: only the addressing differs)."""
    from mpi4py_fft_amd import fftw, asdevice, zeros
    n = shape[axis]
    A = O.rng_array(shape, dt, 5)
    a = asdevice(A)
    for planner, kind in ((fftw.fftn, -1), (fftw.ifftn, 1)):
        nat = planner(a, axes=(axis,), output_array=zeros(shape, dt))
        want = np.asarray(nat.execute_scaled(a, nat.output_array, 0.5)).copy()
        for p in (2, 4, 8):
            ok_len = n % p == 0 and p <= (8 if (n & (n - 1)) == 0 and n >= 32 else 4)
            plan = planner(a, axes=(axis,), output_array=zeros(shape, dt))
            assert plan.set_split(1, p) == ok_len, (shape, axis, p)
            if ok_len:
                got = np.asarray(plan.execute_scaled(a, plan.output_array, 0.5)).reshape(-1)
                assert np.array_equal(got, _packed(want, axis, p)), (shape, axis, kind, p, 'store')
                assert plan.set_split(1, 1)
                assert plan.set_split(0, p)
                src = asdevice(_packed(A, axis, p).reshape(shape))
                got = np.asarray(plan.execute_scaled(src, plan.output_array, 0.5))
                assert np.array_equal(got, want), (shape, axis, kind, p, 'load')
                # both sides at once
                assert plan.set_split(1, p)
                got = np.asarray(plan.execute_scaled(src, plan.output_array, 0.5)).reshape(-1)
                assert np.array_equal(got, _packed(want, axis, p)), (shape, axis, kind, p, 'both')
            plan.destroy()
        nat.destroy()


def test_split_layout_refusals():
    from mpi4py_fft_amd import fftw, asdevice, zeros
    a = asdevice(O.rng_array((8, 16, 32), 'd', 1))
    r = fftw.rfftn(a, axes=(1,))
    assert r.set_split(1, 2) is False                    # real transform along a strided axis
    r2 = fftw.rfftn(a, axes=(2,))
    assert r2.set_split(0, 2) is False                   # the REAL side of a packed-real row plan
    assert r2.set_split(1, 9) is False                   # more than 8 blocks
    r2.destroy()
    c = asdevice(O.rng_array((8, 16, 32), 'D', 1))
    p2 = fftw.fftn(c, axes=(1, 2))
    assert p2.set_split(1, 2) is False                   # two passes
    p3 = fftw.fftn(c, axes=(1,))
    assert p3.set_split(1, 3) is False and p3.set_split(1, 16) is False
    big = asdevice(O.rng_array((2, 8192), 'D', 1))
    p4 = fftw.fftn(big, axes=(1,))
    assert p4.set_split(1, 2) is False                   # four-step
    for q in (r, p2, p3, p4):
        q.destroy()


def test_pinned_staging_round_trip(monkeypatch):
    """np.asarray(u) through the two pinned bounce buffers and the threaded host copy (several
    chunks, odd tail), u[...] = host, and a caller-owned pinned array (host_empty): bytes arrive
    unchanged."""
    from mpi4py_fft_amd import array, empty, host_empty
    monkeypatch.setattr(array, 'PIN_CHUNK_BYTES', 1 << 20)
    monkeypatch.setattr(array, 'PIN_MIN_BYTES', 1 << 16)
    monkeypatch.setattr(array, 'HOST_COPY_THREADS', 3)
    array._pinned.clear()
    rng = np.random.default_rng(3)
    for shape, dt in (((3, 257, 1031), 'D'), ((5, 333, 129), 'f'), ((64, 64), 'd')):
        h = rng.standard_normal(shape).astype(dt)
        if dt == 'D':
            h = h + 1j * rng.standard_normal(shape)
        u = empty(shape, dt)
        u[...] = h
        assert np.array_equal(np.asarray(u), h)
        assert np.array_equal(u.tensor.cpu().numpy(), h)
        p = host_empty(shape, dt)
        p[...] = h
        v = empty(shape, dt)
        v[...] = p
        q = host_empty(shape, dt)
        assert v.get(out=q) is q and np.array_equal(q, h)
        u[...] = h.astype('F' if dt == 'D' else 'd')          # dtype conversion on the way in
        assert np.allclose(np.asarray(u), h, rtol=1e-6)
    array._pinned.clear()


def test_staged_read_back_from_several_threads(monkeypatch):
    """np.asarray(u) from several threads of one process at once (thread ranks; a multi-threaded
    host): the pinned bounce buffers are shared, the read-backs must not interleave.  (Found by
    tools/stress.py mid: 8 thread ranks reading 178 MB results back got each other's chunks.)"""
    import threading
    from mpi4py_fft_amd import array, empty
    monkeypatch.setattr(array, 'PIN_CHUNK_BYTES', 1 << 18)
    monkeypatch.setattr(array, 'PIN_MIN_BYTES', 1 << 16)
    array._pinned.clear()
    hosts = [np.random.default_rng(r).standard_normal((37, 129, 65)) for r in range(6)]
    devs = []
    for h in hosts:
        u = empty(h.shape, 'd')
        u[...] = h
        devs.append(u)
    bad = []

    def body(r):
        for _ in range(5):
            if not np.array_equal(np.asarray(devs[r]), hosts[r]):
                bad.append(r)
    th = [threading.Thread(target=body, args=(r,)) for r in range(len(hosts))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    array._pinned.clear()
    assert not bad, bad


@pytest.mark.parametrize('dt', ['d', 'f'])
@pytest.mark.parametrize('n', [32, 64, 1024, 96, 40])
def test_packed_real_rows_address_uneven_exchange_blocks(n, dt):
    """gfft_plan_set_split on packed-real row plans: the r2c kernel writes, and the c2r kernel reads,
    the half spectrum (n/2 + 1 entries, never an even split) as the all-to-all buffer gfft_pack
    produces for p = 2 ... 8 ranks -- bit for bit."""
    from mpi4py_fft_amd import fftw, asdevice, zeros, _lib
    shape = (6, 5, n)
    nh = n // 2 + 1
    cdt = 'D' if dt == 'd' else 'F'
    a = asdevice(O.rng_array(shape, dt, 21))
    nat = fftw.rfftn(a, axes=(2,))
    want = nat.execute_scaled(a, nat.output_array, 0.25)
    eng = _lib.engine()
    for p in (2, 3, 4, 5, 7, 8):
        packed_want = zeros((6, 5, nh), cdt)
        eng.pack(want.tensor, packed_want.tensor, (6, 5, nh), 2, p, np.dtype(cdt).itemsize)
        plan = fftw.rfftn(a, axes=(2,), output_array=zeros((6, 5, nh), cdt))
        assert plan.set_split(1, p)
        got = np.asarray(plan.execute_scaled(a, plan.output_array, 0.25))
        # (two instantiations of the kernel -- natural and uneven-block store --: the compiler may contract
        # multiply-adds differently, so values agree to the last bit or two, positions exactly)
        pw = np.asarray(packed_want)
        assert np.abs(got - pw).max() <= (1e-14 if dt == 'd' else 1e-6) * np.abs(pw).max(), (n, dt, p, 'r2c store')
        plan.destroy()
        # c2r reading the same buffer gives what it gives on the natural half spectrum
        back_nat = fftw.irfftn(want, s=(n,), axes=(2,), output_array=zeros(shape, dt))
        ref = np.asarray(back_nat.execute_scaled(want, back_nat.output_array, 1.0)).copy()
        back = fftw.irfftn(packed_want, s=(n,), axes=(2,), output_array=zeros(shape, dt))
        assert back.set_split(0, p)
        got = np.asarray(back.execute_scaled(packed_want, back.output_array, 1.0))
        # (two instantiations of the kernel: the compiler may contract multiply-adds differently)
        assert np.abs(got - ref).max() <= (1e-13 if dt == 'd' else 1e-5) * np.abs(ref).max(), (n, dt, p, 'c2r load')
        back.destroy()
        back_nat.destroy()
    nat.destroy()
