"""The two LOCAL stages of a slab-decomposed transform as one launch (gfft_plan_create_guru2, PFFT._fuse_pairs,
pipeline._PairStage).  The reference joins those stages with a whole-array self-Alltoallw
(/root/reference/mpi4py_fft/mpifft.py:324-331, pencil.py:168-183); the parity pins are the same as for every other
path -- the oracle (numpy restatement of the reference) on the same seeded input, the round trip, and the agreement of
the launch forms with one another."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from tests import cases
from oracle import pfft_oracle as O

EPS = {'D': 2.3e-16, 'F': 1.2e-7}


def _rounding(dt, nelem):
    # a transform of nelem points accumulates ~log2(nelem) roundings per value (tests/test_gpu_serial._check)
    return 4 * EPS[dt] * np.log2(nelem)


@pytest.fixture
def small_ring():
    """A hand-off ring of 4 planes, the producer 2 ahead: launches of 8 planes already fuse (the automatic ring wants
    ~200 MiB of planes, plan.cpp fused2_ring), so that test arrays stay oracle-sized."""
    from mpi4py_fft_amd import _lib
    _lib.set_option('fuse2_ring', 4)
    _lib.set_option('fuse2_lag', 2)
    yield
    _lib.set_option('fuse2_ring', 0)
    _lib.set_option('fuse2_lag', 0)


@pytest.mark.parametrize('dt,n,planes,blocks,pitch,launches', [
    ('D', 1024, 16, 1, 0, 1), ('D', 1024, 16, 2, 0, 1), ('D', 1024, 20, 8, 48, 1), ('D', 512, 24, 4, 16, 1), ('D', 512, 16, 1, 0, 1),
    ('F', 1024, 16, 2, 0, 1), ('F', 1024, 24, 8, 32, 1), ('F', 512, 16, 2, 0, 1), ('F', 512, 24, 8, 16, 1),
    ('D', 256, 16, 4, 16, 2), ('F', 256, 16, 2, 0, 2), ('D', 1024, 6, 2, 0, 2),       # no fused kernels / too few planes: two launches
])
@pytest.mark.parametrize('order', ['cols-first', 'rows-first'])
def test_guru2_plans_against_numpy(dt, n, planes, blocks, pitch, launches, order, small_ring):
    """Both directions of a batched 2-D plan whose strided axis is cut into `blocks` blocks on the buffer side, the
    buffer's planes `pitch` elements further apart than their data: forward natural -> buffer, backward buffer ->
    natural, against numpy's fft2 on the same planes."""
    import torch
    from mpi4py_fft_amd import _lib
    eng = _lib.engine()
    prec = 8 if dt == 'D' else 4
    cdt = torch.complex128 if dt == 'D' else torch.complex64
    rng = np.random.default_rng(7)
    x = (rng.standard_normal((planes, n, n)) + 1j * rng.standard_normal((planes, n, n))).astype(dt)
    nb = n // blocks
    E = nb * n + pitch                       # one plane of one block
    bstride = planes * E
    cf = order == 'cols-first'
    hf = eng.plan_create_guru2(prec, -1, (n, n, n), (n, 1, 1), (planes, n * n, E), cf, 1, 0, blocks, bstride)
    hb = eng.plan_create_guru2(prec, +1, (n, n, n), (n, 1, 1), (planes, E, n * n), cf, blocks, bstride, 1, 0)
    assert hf is not None and hb is not None
    if dt == 'F' and n == 512 and not cf:
        launches = 2                         # (the complex64 n = 512 pair exists as [strided -> rows] only)
    assert eng.plan_cost(hf)[2] == launches and eng.plan_cost(hb)[2] == launches, (eng.plan_describe(hf), eng.plan_describe(hb))
    a = torch.from_numpy(x).cuda()
    buf = torch.full((blocks * bstride,), float('nan'), dtype=cdt, device='cuda')
    eng.execute_ptr(hf, a.data_ptr(), buf.data_ptr(), 1.0 / (n * n))
    torch.cuda.synchronize()
    _lib.check_async()
    got = buf.cpu().numpy().reshape(blocks, planes, E)
    want = np.fft.fft2(x.astype('D'), axes=(1, 2)) / (n * n)
    tol = _rounding(dt, n * n)
    for j in range(blocks):
        blk = got[j, :, :nb * n].reshape(planes, nb, n)
        err = np.abs(blk - want[:, j * nb:(j + 1) * nb]).max() / np.abs(want).max()
        assert err <= tol, (j, err, tol)
        if pitch:
            assert np.isnan(got[j, :, nb * n:].real).all()              # the padding between planes is never written
    assert np.array_equal(a.cpu().numpy(), x)                           # input preserved
    back = torch.full((planes, n, n), float('nan'), dtype=cdt, device='cuda')
    eng.execute_ptr(hb, buf.data_ptr(), back.data_ptr(), 1.0)
    torch.cuda.synchronize()
    _lib.check_async()
    rt = np.linalg.norm(back.cpu().numpy() - x) / np.linalg.norm(x)
    assert rt <= tol, (rt, tol)
    eng.plan_destroy(hf)
    eng.plan_destroy(hb)


@pytest.mark.parametrize('n1,n2,planes,blocks,pitch', [(512, 1024, 16, 1, 0), (512, 1024, 20, 4, 32), (1024, 512, 16, 2, 0),
                                                      (1024, 512, 24, 8, 16)])
def test_guru2_unequal_planes(n1, n2, planes, blocks, pitch, small_ring):
    """Planes of n1 x n2 points with n1 != n2 (non-cubic grids): the [strided n1 -> rows n2] pair of either direction as
    one launch, blocks of the strided axis on the buffer side -- against numpy's fft2 and the round trip."""
    import torch
    from mpi4py_fft_amd import _lib
    eng = _lib.engine()
    rng = np.random.default_rng(11)
    x = rng.standard_normal((planes, n1, n2)) + 1j * rng.standard_normal((planes, n1, n2))
    nb = n1 // blocks
    E = nb * n2 + pitch
    bstride = planes * E
    hf = eng.plan_create_guru2(8, -1, (n1, n2, n2), (n2, 1, 1), (planes, n1 * n2, E), True, 1, 0, blocks, bstride)
    hb = eng.plan_create_guru2(8, +1, (n1, n2, n2), (n2, 1, 1), (planes, E, n1 * n2), True, blocks, bstride, 1, 0)
    assert hf is not None and hb is not None
    assert eng.plan_cost(hf)[2] == 1 and eng.plan_cost(hb)[2] == 1, (eng.plan_describe(hf), eng.plan_describe(hb))
    a = torch.from_numpy(x).cuda()
    buf = torch.full((blocks * bstride,), float('nan'), dtype=torch.complex128, device='cuda')
    eng.execute_ptr(hf, a.data_ptr(), buf.data_ptr(), 1.0 / (n1 * n2))
    torch.cuda.synchronize()
    _lib.check_async()
    got = buf.cpu().numpy().reshape(blocks, planes, E)
    want = np.fft.fft2(x, axes=(1, 2)) / (n1 * n2)
    tol = cases.rounding_tol('D', n1 * n2)
    for j in range(blocks):
        blk = got[j, :, :nb * n2].reshape(planes, nb, n2)
        err = np.abs(blk - want[:, j * nb:(j + 1) * nb]).max() / np.abs(want).max()
        assert err <= tol, (j, err, tol)
        if pitch:
            assert np.isnan(got[j, :, nb * n2:].real).all()
    back = torch.full((planes, n1, n2), float('nan'), dtype=torch.complex128, device='cuda')
    eng.execute_ptr(hb, buf.data_ptr(), back.data_ptr(), 1.0)
    torch.cuda.synchronize()
    _lib.check_async()
    rt = np.linalg.norm(back.cpu().numpy() - x) / np.linalg.norm(x)
    assert rt <= tol, (rt, tol)
    # the same plans as two stand-alone launches: the two forms agree to rounding
    _lib.set_option('fuse2_mixed', 0)
    try:
        h2 = eng.plan_create_guru2(8, -1, (n1, n2, n2), (n2, 1, 1), (planes, n1 * n2, E), True, 1, 0, blocks, bstride)
        assert eng.plan_cost(h2)[2] == 2
        buf2 = torch.full((blocks * bstride,), float('nan'), dtype=torch.complex128, device='cuda')
        eng.execute_ptr(h2, a.data_ptr(), buf2.data_ptr(), 1.0 / (n1 * n2))
        torch.cuda.synchronize()
        g2 = buf2.cpu().numpy().reshape(blocks, planes, E)[:, :, :nb * n2]
        assert np.abs(g2 - got[:, :, :nb * n2]).max() / np.abs(want).max() <= tol
        eng.plan_destroy(h2)
    finally:
        _lib.set_option('fuse2_mixed', 1)
    eng.plan_destroy(hf)
    eng.plan_destroy(hb)


@pytest.mark.parametrize('shape', [(512, 16, 1024), (1024, 20, 512)])
def test_one_rank_3d_with_unequal_axes_fuses_its_last_two_passes(shape, small_ring):
    """fftn of a non-cubic array on one rank: axis 1 alone, then [axis 0 -> rows of axis 2] as one launch although
    n0 != n2 -- against numpy."""
    import torch
    from mpi4py_fft_amd import _lib
    eng = _lib.engine()
    rng = np.random.default_rng(12)
    x = rng.standard_normal(shape) + 1j * rng.standard_normal(shape)
    want = np.fft.fftn(x)
    tol = cases.rounding_tol('D', x.size)
    for kind, ref in ((-1, want), (+1, np.conj(np.fft.fftn(np.conj(x))))):
        h = eng.plan_create(list(shape), list(shape), [0, 1, 2], kind, 8)
        desc = eng.plan_describe(h)
        assert eng.plan_cost(h)[2] == 2 and 'fused pair' in desc, desc
        a = torch.from_numpy(x).cuda()
        out = torch.empty_like(a)
        eng.execute_ptr(h, a.data_ptr(), out.data_ptr(), 1.0)
        torch.cuda.synchronize()
        _lib.check_async()
        err = np.abs(out.cpu().numpy() - ref).max() / np.abs(ref).max()
        assert err <= tol, (kind, err, tol)
        eng.plan_destroy(h)


def test_guru2_refusals():
    from mpi4py_fft_amd import _lib
    eng = _lib.engine()
    n = 512
    # a strided row axis, a real kind, blocks on both sides, a block count that does not fit, a length without kernels
    assert eng.plan_create_guru2(8, -1, (n, n, n), (n, 2, 1), (4, n * n, n * n)) is None
    assert eng.plan_create_guru2(8, -2, (n, n, n), (n, 1, 1), (4, n * n, n * n)) is None        # (kind: complex-to-complex only)
    with pytest.raises(_lib.GfftError):
        eng.plan_create_guru2(8, -1, (n, n, n), (n, 1, 1), (0, n * n, n * n))
    assert eng.plan_create_guru2(8, -1, (n, n, n), (n, 1, 1), (4, n * n, n * n), False, 2, n * n, 2, n * n) is None
    assert eng.plan_create_guru2(8, -1, (n, n, n), (n, 1, 1), (4, n * n, n * n), False, 1, 0, 16, n * n) is None
    assert eng.plan_create_guru2(8, -1, (521, n, n), (n, 1, 1), (4, 521 * n, 521 * n)) is None


@pytest.fixture(scope='module')
def fake_rccl():
    import os
    import subprocess
    here = os.path.dirname(os.path.abspath(__file__))
    src, so = os.path.join(here, 'fake_rccl', 'fake_rccl.cpp'), os.path.join(here, 'fake_rccl', 'libfake_rccl.so')
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(['/opt/rocm/bin/hipcc', '-O2', '-std=c++17', '-fPIC', '-shared', '-x', 'hip',
                               '--offload-arch=gfx950', src, '-o', so])
    from mpi4py_fft_amd import _lib
    _lib.check_wire(_lib.lib().gfft_rccl_load(so.encode()))
    yield
    _lib.lib().gfft_rccl_load(None)


@pytest.mark.parametrize('P,shape,dt', [(2, (32, 512, 512), 'D'), (4, (64, 512, 512), 'D'), (8, (128, 512, 512), 'D'),
                                        (2, (32, 1024, 1024), 'D'), (2, (32, 1024, 1024), 'F'),
                                        (2, (32, 512, 1024), 'D'), (4, (64, 1024, 512), 'D'),           # (unequal planes)
                                        (2, (32, 512, 512), 'F')])
def test_slab_grid_runs_its_local_stages_as_one_launch(P, shape, dt, small_ring, fake_rccl, monkeypatch):
    """PFFT on a slab grid: staged with the pair, staged stage by stage (fuse_pairs=False), pipelined with the pair per
    chunk of planes -- against the oracle, each other and the round trip."""
    from mpi4py_fft_amd import PFFT, newDistArray, pipeline
    monkeypatch.setattr(pipeline.Pipeline, 'MIN_CHUNK_BYTES', 0)
    G = O.rng_array(shape, dt, 23)

    def body(comm):
        kw = dict(dtype=dt, grid=[P, 1, 1], exchange='direct')
        plain = PFFT(comm, shape, wire='torch', fuse_pairs=False, **kw)
        staged = PFFT(comm, shape, wire='torch', **kw)
        piped = PFFT(comm, shape, wire='native', **kw)
        assert not plain.forward._pairs and plain.pipeline is None and staged.pipeline is None
        assert list(staged.forward._pairs) == [0] and list(staged.backward._pairs) == [1], 'the pair was not planned'
        assert len(staged.xfftn) == len(staged.axes) == 3
        assert piped.pipeline is not None and piped.pipeline.layout == 'slab-pair', piped.pipeline and piped.pipeline.describe()
        info = piped.pipeline.describe()
        labels = [t[0] for t in staged.forward.stage_times()]
        u = newDistArray(staged, False)
        u[...] = G[staged.local_slice(False)]
        keep = np.asarray(u).copy()
        a0 = np.asarray(plain.forward(u)).copy()
        a = np.asarray(staged.forward(u)).copy()
        b = np.asarray(piped.forward(u)).copy()
        out = newDistArray(piped, True)
        piped.forward(u, out)                          # caller's arrays read / written directly
        c = np.asarray(out).copy()
        assert np.array_equal(np.asarray(u), keep)
        r0 = np.asarray(plain.backward()).copy()
        ra = np.asarray(staged.backward()).copy()
        rb = np.asarray(piped.backward()).copy()
        back = newDistArray(piped, False)
        piped.backward(out, back, normalize=True)
        rn = np.asarray(back).copy()
        sl = staged.local_slice(False)
        for f in (plain, staged, piped):
            f.destroy()
        return a0, a, b, c, r0, ra, rb, rn, sl, info, labels
    res = cases.run_ranks(P, body)
    ref = O.OPFFT(P, shape, dtype=dt, grid=[P, 1, 1])
    want = ref.forward(ref.scatter(G))
    nelem = float(np.prod(shape))
    tol = _rounding(dt, nelem)
    for r, (a0, a, b, c, r0, ra, rb, rn, sl, info, labels) in enumerate(res):
        assert 'one launch' in labels[0], labels
        assert info[0]['chunks'] > 1, info
        assert np.array_equal(a, b) and np.array_equal(a, c), 'pipelined pair differs from the staged pair'
        assert np.array_equal(ra, rb)
        scale = np.abs(want[r]).max()
        assert np.abs(a - want[r]).max() <= tol * scale, (np.abs(a - want[r]).max() / scale, tol)
        assert np.abs(a0 - want[r]).max() <= tol * scale
        assert np.abs(a - a0).max() <= tol * scale                          # one launch against two: rounding only
        assert np.linalg.norm(ra - G[sl]) <= tol * np.linalg.norm(G[sl])
        assert np.linalg.norm(r0 - G[sl]) <= tol * np.linalg.norm(G[sl])
        assert np.allclose(rn * nelem, rb, rtol=1e-5 if dt == 'F' else 1e-12, atol=0)


def test_a_voided_slab_pair_falls_back_to_its_two_passes(small_ring):
    """A guru2 launch that gives up a wait (fuse2_wait_ms = 0: every wait gives up at once) is reported once as GFFT_ERR_VOIDED and the
    plan runs [strided IN -> OUT, rows in place on OUT] from then on -- with the blocks of the all-to-all buffer on either side."""
    import torch
    from mpi4py_fft_amd import _lib
    eng = _lib.engine()
    n, planes, blocks = 1024, 16, 4
    rng = np.random.default_rng(3)
    x = (rng.standard_normal((planes, n, n)) + 1j * rng.standard_normal((planes, n, n))).astype('D')
    nb = n // blocks
    E = nb * n + 16
    bstride = planes * E
    want = np.fft.fft2(x, axes=(1, 2))
    a = torch.from_numpy(x).cuda()
    buf = torch.zeros(blocks * bstride, dtype=torch.complex128, device='cuda')
    back = torch.zeros((planes, n, n), dtype=torch.complex128, device='cuda')
    hf = eng.plan_create_guru2(8, -1, (n, n, n), (n, 1, 1), (planes, n * n, E), True, 1, 0, blocks, bstride)
    hb = eng.plan_create_guru2(8, +1, (n, n, n), (n, 1, 1), (planes, E, n * n), True, blocks, bstride, 1, 0)
    try:
        for h, src, dst in ((hf, a, buf), (hb, buf, back)):
            assert eng.plan_cost(h)[2] == 1
            _lib.set_option('fuse2_wait_ms', 0)
            eng.execute_ptr(h, src.data_ptr(), dst.data_ptr(), 1.0)
            torch.cuda.synchronize()
            _lib.set_option('fuse2_wait_ms', 2000)
            with pytest.raises(RuntimeError, match='gave up'):
                eng.plan_status(h)
            assert 'two stand-alone passes' in eng.plan_describe(h)
            eng.execute_ptr(h, src.data_ptr(), dst.data_ptr(), 1.0 if h is hf else 1.0 / (n * n))       # ... and right
            torch.cuda.synchronize()
            _lib.check_async()
        got = buf.cpu().numpy().reshape(blocks, planes, E)
        for j in range(blocks):
            blk = got[j, :, :nb * n].reshape(planes, nb, n)
            assert np.abs(blk - want[:, j * nb:(j + 1) * nb]).max() <= _rounding('D', n * n) * np.abs(want).max()
        assert np.linalg.norm(back.cpu().numpy() - x) <= _rounding('D', n * n) * np.linalg.norm(x)
    finally:
        _lib.set_option('fuse2_wait_ms', 2000)
        eng.plan_destroy(hf)
        eng.plan_destroy(hb)


def test_pairs_without_packed_exchange_buffers(small_ring, monkeypatch):
    """fuse_pack=False: the pair writes / reads the natural stage array and the Transfer packs with its own kernels."""
    from mpi4py_fft_amd import PFFT
    shape, P = (32, 512, 512), 2

    def body(comm):
        fft = PFFT(comm, shape, dtype='D', grid=[P, 1, 1], wire='torch', exchange='direct', fuse_pack=False)
        out = (list(fft.forward._pairs), [(t.packedA, t.packedB) for t in fft.transfer])
        fft.destroy()
        return out
    for pairs, packed in cases.run_ranks(P, body):
        assert pairs == [0] and packed == [(False, False), (False, False)]
    monkeypatch.setenv('GFFT_FUSE_PACK', '0')
    cases.check_pfft_vs_oracle(P, shape, 'D', grid=[P, 1, 1])
