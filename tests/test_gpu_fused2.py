"""Two dependent passes in ONE persistent launch, the plane between them handed over through the Infinity
Cache (csrc/fft_pow2_impl.h fft_fused2_kernel, csrc/fft_fused_f64.hip): passes 1 + 2 (forward) / 2 + 3
(backward) of the single-GPU 3-D schedule at n0 = n2 = 1024 in fp64, and both passes of a four-step transform
of length 2^20 (BASELINE config C2).  Checked against the unfused plans of the same library (same arithmetic
per line), against numpy, and for run-to-run determinism under every ring / lag setting -- a hand-off that
raced (a tile read before its plane was complete, a slot overwritten too early) would show as a difference."""
import numpy as np
import pytest

from tests import cases

pytestmark = pytest.mark.gpu


def _opts(**kw):
    from mpi4py_fft_amd import _lib
    for k, v in kw.items():
        _lib.set_option(k, v)


@pytest.fixture(autouse=True)
def _restore():
    yield
    _opts(fuse2=1, fuse2_ring=0, fuse2_lag=0, fuse2_kinds=510, fuse2_wait_ms=2000, fuse2_f32=1, wtile=1)


def _plans(shape, axes, fuse, ring=8, lag=4, kinds=15, dt='D'):
    from mpi4py_fft_amd import fftw, zeros
    _opts(fuse2=fuse, fuse2_ring=ring, fuse2_lag=lag, fuse2_kinds=kinds)
    a = zeros(shape, dt)
    f = fftw.fftn(a, axes=axes)
    b = fftw.ifftn(f.output_array, axes=axes, output_array=zeros(shape, dt))
    return a, f, b


# (axis 1 a power of two: the tile-major workspace W[i0][tile of 16 columns][k1][16] under the pair [axis 0 -> rows], plan_fused3 option
# wtile; 40: pitched rows, and the tile-major layout forced with wtile = 2)
@pytest.mark.parametrize('shape,dt,wtile', [((1024, 16, 1024), 'D', 1), ((1024, 40, 1024), 'D', 1), ((1024, 64, 1024), 'D', 1),
                                            ((1024, 40, 1024), 'D', 2), ((1024, 64, 1024), 'D', 0)])
def test_fused_3d_schedule_matches_the_unfused_one(shape, dt, wtile):
    from mpi4py_fft_amd import _lib
    _opts(wtile=wtile)
    tiled = wtile == 2 or (wtile == 1 and shape[1] & (shape[1] - 1) == 0)
    rng = np.random.default_rng(5)
    x = (rng.standard_normal(shape) + 1j * rng.standard_normal(shape)).astype(dt)
    eps = 1e-13 if dt == 'D' else 1e-5
    a0, f0, b0 = _plans(shape, (0, 1, 2), 0, dt=dt)
    assert 'fused pair' not in _lib.engine().plan_describe(f0._plan)
    a0[...] = x
    want = np.asarray(f0.execute_scaled(a0, f0.output_array, 1.0)).copy()
    wantb = np.asarray(b0.execute_scaled(f0.output_array, b0.output_array, 1.0 / x.size)).copy()
    ref = np.fft.fftn(x.astype('D'))
    assert np.abs(want - ref).max() <= cases.tol_for(dt, ref.size) * np.abs(ref).max()
    for f in (f0, b0):
        f.destroy()
    # kinds 15: both directions run axis 1, then the fused pair [axis 0 -> rows]; kinds 1: only the mirror pair
    # [rows -> axis 0] exists, which the forward direction then takes (the backward one stays unfused)
    for ring, lag, kinds in ((8, 4, 15), (8, 1, 15), (5, 4, 15), (3, 2, 15), (16, 8, 15), (8, 4, 1), (4, 3, 1)):
        if shape[1] < 2 * ring:
            continue
        a1, f1, b1 = _plans(shape, (0, 1, 2), 1, ring, lag, kinds, dt)
        desc = _lib.engine().plan_describe(f1._plan)
        assert ('fused pair (strided -> rows)' if kinds == 15 else 'fused pair (rows -> strided)') in desc, desc
        assert 'ring of %d slots' % ring in desc, desc
        assert ('tile-major workspace, tiles of 16 columns' in desc) == (tiled and kinds == 15), desc
        assert ('fused pair (strided -> rows)' in _lib.engine().plan_describe(b1._plan)) == (kinds == 15)
        a1[...] = x
        for rep in range(3):
            got = np.asarray(f1.execute_scaled(a1, f1.output_array, 1.0))
            assert np.abs(got - want).max() <= eps * np.abs(want).max(), (ring, lag, rep)
            if rep == 0:
                first = got.copy()
            assert np.array_equal(got, first), ('forward not reproducible', ring, lag, rep)
            back = np.asarray(b1.execute_scaled(f1.output_array, b1.output_array, 1.0 / x.size))
            assert np.abs(back - wantb).max() <= eps * np.abs(wantb).max(), (ring, lag, rep)
            assert np.abs(back - x).max() <= 10 * eps * np.abs(x).max()
        assert np.array_equal(np.asarray(a1), x)          # out of place: the input is preserved
        f1.destroy()
        b1.destroy()


@pytest.mark.parametrize('n', [960, 896])
def test_fused_3d_pair_on_unequal_width_stage_lengths(n):
    """Round 5: [axis 0 -> rows] of the complex 3-D schedule at n0 = n2 = 960 / 896 (csrc/fft_fused_f64.hip Fused960 / Fused896: the
    tiles' stages keep different numbers of values per thread): against the unfused plans, numpy, run to run."""
    from mpi4py_fft_amd import _lib
    shape = (n, 40, n)
    rng = np.random.default_rng(n)
    x = rng.standard_normal(shape) + 1j * rng.standard_normal(shape)
    a0, f0, b0 = _plans(shape, (0, 1, 2), 0)
    assert 'fused pair' not in _lib.engine().plan_describe(f0._plan)
    a0[...] = x
    want = np.asarray(f0.execute_scaled(a0, f0.output_array, 1.0)).copy()
    ref = np.fft.fftn(x)
    assert np.abs(want - ref).max() <= cases.tol_for('D', ref.size) * np.abs(ref).max()
    f0.destroy()
    b0.destroy()
    for ring, lag in ((8, 4), (12, 6), (5, 2)):
        a1, f1, b1 = _plans(shape, (0, 1, 2), 1, ring, lag, 126)
        assert 'fused pair (strided -> rows)' in _lib.engine().plan_describe(f1._plan), _lib.engine().plan_describe(f1._plan)
        assert 'fused pair (strided -> rows)' in _lib.engine().plan_describe(b1._plan)
        a1[...] = x
        for rep in range(3):
            got = np.asarray(f1.execute_scaled(a1, f1.output_array, 1.0))
            assert np.abs(got - want).max() <= 1e-13 * np.abs(want).max(), (ring, lag, rep)
            if rep == 0:
                first = got.copy()
            assert np.array_equal(got, first)
            back = np.asarray(b1.execute_scaled(f1.output_array, b1.output_array, 1.0 / x.size))
            assert np.abs(back - x).max() <= 1e-12 * np.abs(x).max()
        f1.destroy()
        b1.destroy()


@pytest.mark.parametrize('shape,dt', [((32, 1 << 20), 'D'), ((64, 1 << 20), 'F')])      # (complex64: round 5, csrc/fft_fused_f32.hip)
def test_fused_four_step_matches_the_two_launch_form(shape, dt):
    from mpi4py_fft_amd import _lib
    rng = np.random.default_rng(6)
    x = (rng.standard_normal(shape) + 1j * rng.standard_normal(shape)).astype(dt)
    eps = 1e-13 if dt == 'D' else 1e-5
    a0, f0, b0 = _plans(shape, (1,), 0, dt=dt)
    a0[...] = x
    want = np.asarray(f0.execute_scaled(a0, f0.output_array, 1.0)).copy()
    ref = np.fft.fft(x[:2].astype('D'), axis=1)
    assert np.abs(want[:2] - ref).max() <= cases.tol_for(dt, ref.size) * np.abs(ref).max()
    f0.destroy()
    b0.destroy()
    # kinds 15: [strided + transposing store] -> [strided]; 31: [strided] -> [rows + transposing store] (the default)
    for ring, lag, kinds in ((8, 4, 15), (4, 1, 15), (16, 8, 15), (8, 4, 31), (4, 1, 31), (12, 6, 31), (0, 0, 31)):
        a1, f1, b1 = _plans(shape, (1,), 1, ring, lag, kinds, dt=dt)
        # (complex64 has the [strided -> strided] form only: measured ahead of the other one, csrc/fft_fused_f32.hip)
        assert ('fused pair (four-step)' if (kinds == 15 or dt == 'F') else 'fused pair (four-step: strided -> rows') in _lib.engine().plan_describe(f1._plan)
        a1[...] = x
        for rep in range(3):
            got = np.asarray(f1.execute_scaled(a1, f1.output_array, 1.0))
            assert np.abs(got - want).max() <= eps * np.abs(want).max(), (ring, lag, kinds, rep)
            back = np.asarray(b1.execute_scaled(f1.output_array, b1.output_array, 1.0 / shape[1]))
            assert np.abs(back - x).max() <= 10 * eps * np.abs(x).max()
        f1.destroy()
        b1.destroy()


def test_shapes_without_a_paying_pair_keep_the_unfused_plans():
    from mpi4py_fft_amd import _lib
    # fewer planes than two rings: nothing to pipeline; fp32 four-step and n = 512: measured slower fused (fft_fused_f64.hip)
    for shape, axes, dt in (((4, 1 << 20), (1,), 'D'), ((32, 1 << 20), (1,), 'F'), ((512, 32, 512), (0, 1, 2), 'D')):
        a, f, b = _plans(shape, axes, 1, 0, 0, 126, dt=dt)          # (ring / lag automatic)
        assert 'fused pair' not in _lib.engine().plan_describe(f._plan), (shape, dt)
        f.destroy()
        b.destroy()


@pytest.mark.parametrize('dt', ['D', 'F'])
def test_fused_batched_2d_transform(dt):
    """fftn over the last two axes of a 3-D array (the leading stage of a slab-decomposed PFFT with
    collapse=True): [columns] -> [rows] plane by plane in one launch (strided reads, whole rows written: round 6)."""
    from mpi4py_fft_amd import _lib
    shape = (24, 1024, 1024)
    rng = np.random.default_rng(8)
    x = (rng.standard_normal(shape) + 1j * rng.standard_normal(shape)).astype(dt)
    eps = 1e-13 if dt == 'D' else 1e-5
    a0, f0, b0 = _plans(shape, (1, 2), 0, dt=dt)
    a0[...] = x
    want = np.asarray(f0.execute_scaled(a0, f0.output_array, 1.0)).copy()
    ref = np.fft.fftn(x[:3].astype('D'), axes=(1, 2))
    assert np.abs(want[:3] - ref).max() <= cases.tol_for(dt, ref.size) * np.abs(ref).max()
    f0.destroy()
    b0.destroy()
    for ring, lag in ((8, 4), (4, 2), (12, 6)):
        if shape[0] < 2 * ring:
            continue
        a1, f1, b1 = _plans(shape, (1, 2), 1, ring, lag, dt=dt)
        assert 'fused pair (strided -> rows)' in _lib.engine().plan_describe(f1._plan)
        a1[...] = x
        for rep in range(3):
            got = np.asarray(f1.execute_scaled(a1, f1.output_array, 1.0))
            assert np.abs(got - want).max() <= eps * np.abs(want).max(), (ring, lag, rep)
            back = np.asarray(b1.execute_scaled(f1.output_array, b1.output_array, 1.0 / (1024 * 1024)))
            assert np.abs(back - x).max() <= 10 * eps * np.abs(x).max()
        f1.destroy()
        b1.destroy()


def test_fused_launch_replays_from_a_hip_graph():
    """A fused launch is a memset of its counters, the persistent kernel and the one-thread check kernel: all
    three capture into a HIP graph (after one warm-up execution, which allocates the ring) and replay."""
    import torch
    from mpi4py_fft_amd import _lib
    shape = (1024, 16, 1024)
    rng = np.random.default_rng(9)
    x = rng.standard_normal(shape) + 1j * rng.standard_normal(shape)
    a, f, b = _plans(shape, (0, 1, 2), 1)
    assert 'fused pair' in _lib.engine().plan_describe(f._plan)
    a[...] = x
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        want = f.execute_scaled(a, f.output_array, 1.0).tensor.clone()         # warm-up on the capture stream
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            f.execute_scaled(a, f.output_array, 1.0)
        for rep in range(3):
            f.output_array.tensor.zero_()
            g.replay()
            torch.cuda.synchronize()
            assert torch.equal(f.output_array.tensor, want), rep
    f.destroy()
    b.destroy()


@pytest.mark.parametrize('shape,axes', [((1024, 16, 1024), (0, 1, 2)), ((24, 1024, 1024), (1, 2)), ((32, 1 << 20), (1,))])
def test_fused_plans_execute_in_place(shape, axes):
    """d_in == d_out (allowed for complex plans, gfft.h): a plane's input is consumed in full -- its last A tile
    stored -- before the first B tile of that plane writes."""
    from mpi4py_fft_amd import fftw, zeros, _lib
    rng = np.random.default_rng(10)
    x = rng.standard_normal(shape) + 1j * rng.standard_normal(shape)
    _opts(fuse2=0)
    a = zeros(shape, 'D')
    a[...] = x
    f0 = fftw.fftn(a, axes=axes)
    want = np.asarray(f0.execute_scaled(a, f0.output_array, 1.0)).copy()
    f0.destroy()
    _opts(fuse2=1, fuse2_kinds=15)
    f1 = fftw.fftn(a, axes=axes, output_array=a)
    assert 'fused pair' in _lib.engine().plan_describe(f1._plan)
    for rep in range(2):
        a[...] = x
        got = np.asarray(f1.execute_scaled(a, a, 1.0))
        assert np.abs(got - want).max() <= 1e-13 * np.abs(want).max(), rep
    f1.destroy()


@pytest.mark.parametrize('shape,axes', [((1024, 40, 1024), (0, 1, 2)), ((1024, 64, 1024), (0, 1, 2)), ((32, 1 << 20), (1,))])
def test_a_fused_launch_that_gives_up_a_wait_is_an_error_and_the_plan_recovers(shape, axes):
    """A wait inside a fused launch that outlasts `fuse2_wait_ms` (a device shared with a long foreign kernel, a
    debugger) voids the launch: no hang, no trap that would poison the HIP context and every other plan of the
    process.  The library reports it from its next call as a RuntimeError -- the reference's error contract,
    mpi4py_fft/fftw/fftw_xfftn.pyx:152-153 -- and the SAME plan then runs the pair as stand-alone passes."""
    import torch
    from mpi4py_fft_amd import _lib
    rng = np.random.default_rng(9)
    x = rng.standard_normal(shape) + 1j * rng.standard_normal(shape)
    ref = np.fft.fftn(x, axes=axes)
    a, f, b = _plans(shape, axes, 1, 8, 4, 31)
    assert 'fused pair' in _lib.engine().plan_describe(f._plan)
    # ((1024, 64, 1024): the stand-alone forms then run on the tile-major workspace the plan was laid out for)
    assert ('tile-major workspace' in _lib.engine().plan_describe(f._plan)) == (shape == (1024, 64, 1024))
    a[...] = x
    good = np.asarray(f.execute_scaled(a, f.output_array, 1.0)).copy()       # the healthy launch first
    assert np.abs(good - ref).max() <= cases.tol_for('D', ref.size) * np.abs(ref).max()
    _opts(fuse2_wait_ms=0)                         # every wait gives up at once
    f.execute_scaled(a, f.output_array, 1.0)       # enqueues; the launch voids itself on the device
    torch.cuda.synchronize()
    with pytest.raises(RuntimeError, match='gave up'):
        np.asarray(f.output_array)                 # reading results: the failure surfaces here ...
    _lib.check_async()                             # ... once
    _opts(fuse2_wait_ms=2000)
    desc = _lib.engine().plan_describe(f._plan)
    assert 'fused pair' not in desc and 'two stand-alone passes' in desc, desc
    for rep in range(2):
        got = np.asarray(f.execute_scaled(a, f.output_array, 1.0))
        assert np.abs(got - ref).max() <= cases.tol_for('D', ref.size) * np.abs(ref).max()
    # the process, the context and the other plans are alive: the backward plan still runs fused
    assert 'fused pair' in _lib.engine().plan_describe(b._plan)
    back = np.asarray(b.execute_scaled(f.output_array, b.output_array, 1.0 / np.prod([shape[i] for i in axes])))
    assert np.abs(back - x).max() <= 1e-12 * np.abs(x).max()
    f.destroy()
    b.destroy()


def test_a_pending_failure_is_reported_by_the_next_execute():
    import torch
    from mpi4py_fft_amd import _lib
    shape = (1024, 32, 1024)
    a, f, b = _plans(shape, (0, 1, 2), 1, 8, 4, 15)
    _opts(fuse2_wait_ms=0)
    f.execute_scaled(a, f.output_array, 1.0)
    torch.cuda.synchronize()
    _opts(fuse2_wait_ms=2000)
    with pytest.raises(RuntimeError, match='stand-alone launches from now on'):
        f.execute_scaled(a, f.output_array, 1.0)
    f.execute_scaled(a, f.output_array, 1.0)        # and then it just works
    torch.cuda.synchronize()
    _lib.check_async()
    f.destroy()
    b.destroy()


def test_a_voided_launch_stays_with_its_plan_and_two_of_them_are_both_reported():
    """The event belongs to the plan that voided (include/gfft.h gfft_plan_status): another plan's execute is not
    refused by it, scalar reads surface it, and two plans voiding before anybody looks are both named."""
    import torch
    from mpi4py_fft_amd import _lib, PFFT, comm
    shape = (1024, 16, 1024)
    a, f, b = _plans(shape, (0, 1, 2), 1, 8, 4, 21)
    a2, f2, b2 = _plans(shape, (0, 1, 2), 1, 8, 4, 22)
    _opts(fuse2_wait_ms=0)
    f.execute_scaled(a, f.output_array, 1.0)
    f2.execute_scaled(a2, f2.output_array, 1.0)
    torch.cuda.synchronize()
    _opts(fuse2_wait_ms=2000)
    b.execute_scaled(f.output_array, b.output_array, 1.0)        # an unrelated plan: runs, the events stay pending
    torch.cuda.synchronize()
    b.status()
    with pytest.raises(RuntimeError, match='gave up'):
        f.status()
    f.status()                                                   # reported once
    with pytest.raises(RuntimeError, match='gave up'):
        f2.output_array[0, 0, 0]                                 # a scalar read synchronises: the other event surfaces here
    _lib.check_async()
    # PFFT.check(): synchronise and drain (collective on a grid)
    p = PFFT(comm.COMM_SELF, (256, 256, 256), dtype='D')
    p.forward()
    p.check()
    p.destroy()
    for x in (f, b, f2, b2):
        x.destroy()


def test_the_round3_kernel_set_is_still_selectable_and_agrees():
    """fuse2 = 3: the pairs on 16 values per thread / two LDS exchanges (the default, 1, runs 32 values per thread and
    one exchange): same results to rounding, both against numpy."""
    shape = (1024, 16, 1024)
    rng = np.random.default_rng(12)
    x = rng.standard_normal(shape) + 1j * rng.standard_normal(shape)
    ref = np.fft.fftn(x)
    outs = []
    for fuse in (1, 3):
        a, f, b = _plans(shape, (0, 1, 2), fuse, 8, 4, 15)
        from mpi4py_fft_amd import _lib
        assert 'fused pair (strided -> rows)' in _lib.engine().plan_describe(f._plan)
        a[...] = x
        got = np.asarray(f.execute_scaled(a, f.output_array, 1.0))
        assert np.abs(got - ref).max() <= cases.tol_for('D', ref.size) * np.abs(ref).max()
        back = np.asarray(b.execute_scaled(f.output_array, b.output_array, 1.0 / x.size))
        assert np.abs(back - x).max() <= 1e-12 * np.abs(x).max()
        outs.append(got.copy())
        f.destroy()
        b.destroy()
    assert np.abs(outs[0] - outs[1]).max() <= 1e-13 * np.abs(ref).max()


@pytest.mark.parametrize('shape', [(40, 1024, 1024), (1024, 40, 1024), (48, 1024, 2048), (1024, 48, 2048)])
def test_fused_pairs_of_real_transforms(shape):
    """r2c: [packed-real rows -> axis 1] on the contiguous planes i0; c2r: [axis 0 -> packed-real rows] on the planes
    i1 (csrc/fft_fused_real_f64.hip) -- the reference's default dtype is real (mpifft.py:202, fftw/xfftn.py:173-326).
    Against numpy, against the unfused plans of the same library, and reproducible run to run."""
    from mpi4py_fft_amd import _lib, fftw, zeros
    rng = np.random.default_rng(21)
    x = rng.standard_normal(shape)
    ref = np.fft.rfftn(x)
    cshape = ref.shape
    res = {}
    for fuse in (0, 1):
        _opts(fuse2=fuse, fuse2_ring=8, fuse2_lag=4, fuse2_kinds=126)
        a = zeros(shape, 'd')
        f = fftw.rfftn(a, axes=(0, 1, 2))
        c = zeros(cshape, 'D')
        b = fftw.irfftn(c, s=shape, axes=(0, 1, 2))
        df, db = _lib.engine().plan_describe(f._plan), _lib.engine().plan_describe(b._plan)
        if fuse:
            assert ('fused pair (r2c rows -> strided)' in df) == (shape[1] == 1024), df
            # (rows of 2048 reals too since round 5: option c2r_2048, on once the workspace pitch stopped aliasing)
            assert ('fused pair (strided -> c2r rows)' in db) == (shape[0] == 1024), db
        else:
            assert 'fused pair' not in df + db
        a[...] = x
        for rep in range(3):
            got = np.asarray(f.execute_scaled(a, f.output_array, 1.0))
            assert np.abs(got - ref).max() <= cases.tol_for('D', ref.size) * np.abs(ref).max(), (fuse, rep)
            if rep == 0:
                first = got.copy()
            assert np.array_equal(got, first), ('forward not reproducible', fuse, rep)
        c[...] = ref
        for rep in range(3):
            back = np.asarray(b.execute_scaled(c, b.output_array, 1.0 / x.size))
            assert np.abs(back - x).max() <= 1e-12 * np.abs(x).max(), (fuse, rep)
        res[fuse] = (first, back.copy())
        f.destroy()
        b.destroy()
    assert np.abs(res[0][0] - res[1][0]).max() <= 1e-13 * np.abs(ref).max()
    assert np.abs(res[0][1] - res[1][1]).max() <= 1e-13 * np.abs(x).max()


@pytest.mark.parametrize('shape', [(1024, 48, 1024), (1024, 64, 1024)])          # (48 planes: just enough for the ring of 24)
def test_fused_pair_of_the_complex64_schedule(shape):
    """[axis 0 -> rows] of the complex64 3-D schedule (csrc/fft_fused_f32.hip), on a ring sized in bytes: against numpy
    in double precision and against the unfused plans, reproducible.  64 planes: on the tile-major workspace (tiles of 32 columns)."""
    from mpi4py_fft_amd import _lib
    rng = np.random.default_rng(31)
    x = (rng.standard_normal(shape) + 1j * rng.standard_normal(shape)).astype('F')
    ref = np.fft.fftn(x.astype('D'))
    res = {}
    for fuse in (0, 1):
        a, f, b = _plans(shape, (0, 1, 2), fuse, 0, 0, 126, dt='F')
        desc = _lib.engine().plan_describe(f._plan)
        assert ('fused pair (strided -> rows)' in desc) == bool(fuse), desc
        if fuse:
            assert 'ring of 24 slots' in desc, desc          # 8.25 MiB planes: twice the slots of the complex128 pair
            assert ('tile-major workspace, tiles of 32 columns' in desc) == (shape[1] == 64), desc
        a[...] = x
        for rep in range(3):
            got = np.asarray(f.execute_scaled(a, f.output_array, 1.0))
            assert np.abs(got - ref).max() <= cases.tol_for('F', ref.size) * np.abs(ref).max()
            if rep == 0:
                first = got.copy()
            assert np.array_equal(got, first)
        back = np.asarray(b.execute_scaled(f.output_array, b.output_array, 1.0 / x.size))
        assert np.abs(back - x).max() <= 2 * cases.tol_for('F', x.size) * np.abs(x).max()
        res[fuse] = first
        f.destroy()
        b.destroy()
    assert np.abs(res[0] - res[1]).max() <= 1e-5 * np.abs(ref).max()


def test_fused_pair_at_n_512_and_where_it_stays_off():
    """complex128, n0 = n2 = 512: 32 values per thread on 256-thread workgroups, two per CU, planes of 4 MiB on a ring of
    48 -- which needs 96 planes; with fewer the pair would lose (measured) and the plan stays unfused."""
    from mpi4py_fft_amd import _lib
    rng = np.random.default_rng(33)
    for shape, fused in (((512, 96, 512), True), ((512, 64, 512), False)):
        x = rng.standard_normal(shape) + 1j * rng.standard_normal(shape)
        ref = np.fft.fftn(x)
        a, f, b = _plans(shape, (0, 1, 2), 1, 0, 0, 126)
        desc = _lib.engine().plan_describe(f._plan)
        assert ('fused pair (strided -> rows)' in desc) == fused, desc
        if fused:
            assert 'ring of 48 slots' in desc, desc
        a[...] = x
        for rep in range(3):
            got = np.asarray(f.execute_scaled(a, f.output_array, 1.0))
            assert np.abs(got - ref).max() <= cases.tol_for('D', ref.size) * np.abs(ref).max()
            if rep == 0:
                first = got.copy()
            assert np.array_equal(got, first)
        back = np.asarray(b.execute_scaled(f.output_array, b.output_array, 1.0 / x.size))
        assert np.abs(back - x).max() <= 1e-12 * np.abs(x).max()
        f.destroy()
        b.destroy()


@pytest.mark.parametrize('shape', [(96, 1024, 1024), (1024, 96, 1024)])
def test_fused_pairs_of_real_fp32_transforms(shape):
    """The real pairs in single precision (csrc/fft_fused_real_f32.hip), planes of 4 MiB on a ring of 46: r2c on
    (96, 1024, 1024), c2r on (1024, 96, 1024); against numpy in double precision and against the unfused plans."""
    from mpi4py_fft_amd import _lib, fftw, zeros
    rng = np.random.default_rng(41)
    x = rng.standard_normal(shape).astype('f')
    ref = np.fft.rfftn(x.astype('d'))
    res = {}
    for fuse in (0, 1):
        _opts(fuse2=fuse, fuse2_ring=0, fuse2_lag=0, fuse2_kinds=126, fuse2_f32=2)          # (2: the real fp32 pairs are off by default, measured level)
        a = zeros(shape, 'f')
        f = fftw.rfftn(a, axes=(0, 1, 2))
        c = zeros(ref.shape, 'F')
        b = fftw.irfftn(c, s=shape, axes=(0, 1, 2))
        df, db = _lib.engine().plan_describe(f._plan), _lib.engine().plan_describe(b._plan)
        if fuse:
            assert ('fused pair (r2c rows -> strided)' in df) == (shape[1] == 1024), df
            assert ('fused pair (strided -> c2r rows)' in db) == (shape[0] == 1024), db
        a[...] = x
        got = np.asarray(f.execute_scaled(a, f.output_array, 1.0)).copy()
        assert np.abs(got - ref).max() <= cases.tol_for('F', ref.size) * np.abs(ref).max()
        assert np.array_equal(np.asarray(f.execute_scaled(a, f.output_array, 1.0)), got)
        c[...] = ref.astype('F')
        back = np.asarray(b.execute_scaled(c, b.output_array, 1.0 / x.size)).copy()
        assert np.abs(back - x).max() <= 2 * cases.tol_for('F', x.size) * np.abs(x).max()
        res[fuse] = (got, back)
        f.destroy()
        b.destroy()
    assert np.abs(res[0][0] - res[1][0]).max() <= 1e-5 * np.abs(ref).max()
    assert np.abs(res[0][1] - res[1][1]).max() <= 1e-5 * np.abs(x).max()


def test_fused_batched_2d_transform_at_n_512():
    """fftn over the last two axes at n1 = n2 = 512 (the leading stage of the slab-decomposed C3 with collapse=True):
    [rows -> columns] plane by plane on the n = 512 kernels, planes of 4 MiB."""
    from mpi4py_fft_amd import _lib
    shape = (96, 512, 512)
    rng = np.random.default_rng(43)
    x = rng.standard_normal(shape) + 1j * rng.standard_normal(shape)
    ref = np.fft.fftn(x, axes=(1, 2))
    a, f, b = _plans(shape, (1, 2), 1, 0, 0, 126)
    assert 'fused pair (strided -> rows)' in _lib.engine().plan_describe(f._plan)
    a[...] = x
    got = np.asarray(f.execute_scaled(a, f.output_array, 1.0))
    assert np.abs(got - ref).max() <= cases.tol_for('D', ref.size) * np.abs(ref).max()
    back = np.asarray(b.execute_scaled(f.output_array, b.output_array, 1.0 / (512 * 512)))
    assert np.abs(back - x).max() <= 1e-12 * np.abs(x).max()
    f.destroy()
    b.destroy()


def test_recovery_of_the_real_pairs_and_the_2d_pair():
    """The stand-alone form a voided launch falls back to exists for every pair kind: r2c planes, c2r, batched 2-D."""
    import torch
    from mpi4py_fft_amd import _lib, fftw, zeros
    rng = np.random.default_rng(51)
    _opts(fuse2=1, fuse2_ring=0, fuse2_lag=0, fuse2_kinds=126)
    # r2c on (40, 1024, 1024), c2r on (1024, 40, 1024)
    for shape, forward in (((40, 1024, 1024), True), ((1024, 40, 1024), False)):
        x = rng.standard_normal(shape)
        ref = np.fft.rfftn(x)
        if forward:
            a = zeros(shape, 'd')
            p = fftw.rfftn(a, axes=(0, 1, 2))
            a[...] = x
            want, scale = ref, 1.0
        else:
            a = zeros(ref.shape, 'D')
            p = fftw.irfftn(a, s=shape, axes=(0, 1, 2))
            a[...] = ref
            want, scale = x, 1.0 / x.size
        assert 'fused pair' in _lib.engine().plan_describe(p._plan)
        _opts(fuse2_wait_ms=0)
        p.execute_scaled(a, p.output_array, scale)
        torch.cuda.synchronize()
        _opts(fuse2_wait_ms=2000)
        with pytest.raises(RuntimeError, match='gave up'):
            _lib.check_async()
        assert 'two stand-alone passes' in _lib.engine().plan_describe(p._plan)
        if not forward:
            a[...] = ref                      # (multi-axis c2r preserves its input; refreshed all the same)
        got = np.asarray(p.execute_scaled(a, p.output_array, scale))
        assert np.abs(got - want).max() <= cases.tol_for('D', want.size) * np.abs(want).max()
        p.destroy()
    shape = (24, 1024, 1024)
    x = rng.standard_normal(shape) + 1j * rng.standard_normal(shape)
    a, f, b = _plans(shape, (1, 2), 1, 0, 0, 126)
    assert 'fused pair (strided -> rows)' in _lib.engine().plan_describe(f._plan)
    a[...] = x
    _opts(fuse2_wait_ms=0)
    f.execute_scaled(a, f.output_array, 1.0)
    torch.cuda.synchronize()
    _opts(fuse2_wait_ms=2000)
    with pytest.raises(RuntimeError, match='gave up'):
        _lib.check_async()
    got = np.asarray(f.execute_scaled(a, f.output_array, 1.0))
    ref = np.fft.fftn(x, axes=(1, 2))
    assert np.abs(got - ref).max() <= cases.tol_for('D', ref.size) * np.abs(ref).max()
    f.destroy()
    b.destroy()
