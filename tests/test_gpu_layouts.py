"""The exchange-buffer layouts of round 3 at the C ABI (include/gfft.h: gfft_plan_set_tiles,
gfft_plan_set_flat, gfft_plan_set_split_slabs): each is the SAME transform as the natural-layout
plan, bit for bit, with the data in another place -- checked by re-laying the natural plan's result
out on the host."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _host(t):
    return t.cpu().numpy()


def _natural_fft(x, axis, prec):
    """The natural-layout guru plan along `axis` of a 3-D complex array (the bit-level reference)."""
    import torch
    from mpi4py_fft_amd import _lib
    eng = _lib.engine()
    shape = x.shape
    st = [shape[1] * shape[2], shape[2], 1]
    dims = [(shape[d], st[d], st[d]) for d in range(3) if d != axis]
    h = eng.plan_create_guru(prec, -1, (shape[axis], st[axis], st[axis]), dims)
    assert h is not None
    a, b = _dev(x), torch.empty(x.shape, dtype=torch.from_numpy(x).dtype, device='cuda')
    eng.execute_ptr(h, a.data_ptr(), b.data_ptr(), 1.0)
    torch.cuda.synchronize()
    eng.plan_destroy(h)
    return _host(b)


@pytest.mark.parametrize('dt,tile,n', [('D', 16, 256), ('D', 16, 1024), ('F', 32, 512), ('F', 32, 2048), ('F', 16, 256)])
@pytest.mark.parametrize('p', [1, 2, 4])
def test_row_plans_write_and_read_tile_major_lines(dt, tile, n, p):
    import torch
    from mpi4py_fft_amd import _lib
    eng = _lib.engine()
    prec = _lib.precision_of(dt)
    n0, n1 = 3, 8
    rng = np.random.default_rng(5)
    x = (rng.standard_normal((n0, n1, n)) + 1j * rng.standard_normal((n0, n1, n))).astype(dt)
    want = _natural_fft(x, 2, prec)
    w = n // p
    # buffer: [block][slab i0][tile][row i1][tile entries]
    bstride, slab, tS = n0 * n1 * w, n1 * w, n1 * tile
    h = eng.plan_create_guru(prec, -1, (n, 1, 1), [(n0, n1 * n, slab), (n1, n, tile)], 1, 0, p, bstride)
    assert h is not None
    assert eng.plan_set_tiles(h, 1, tile, tS)
    a = _dev(x)
    b = torch.zeros(n0 * n1 * n, dtype=a.dtype, device='cuda')
    eng.execute_ptr(h, a.data_ptr(), b.data_ptr(), 1.0)
    torch.cuda.synchronize()
    got = _host(b).reshape(p, n0, w // tile, n1, tile)
    ref = want.reshape(n0, n1, p, w // tile, tile).transpose(2, 0, 3, 1, 4)
    assert np.array_equal(got, ref)
    # ... and read it back the same way (inverse, unscaled): x * n
    hb = eng.plan_create_guru(prec, +1, (n, 1, 1), [(n0, slab, n1 * n), (n1, tile, n)], p, bstride, 1, 0)
    assert hb is not None and eng.plan_set_tiles(hb, 0, tile, tS)
    c = torch.zeros_like(a)
    eng.execute_ptr(hb, b.data_ptr(), c.data_ptr(), 1.0 / n)
    torch.cuda.synchronize()
    assert np.abs(_host(c) - x).max() <= (1e-12 if dt == 'D' else 2e-5) * np.abs(x).max()
    eng.plan_destroy(h)
    eng.plan_destroy(hb)


def test_row_tiles_are_refused_where_the_thread_layout_does_not_fit():
    from mpi4py_fft_amd import _lib
    eng = _lib.engine()
    h = eng.plan_create_guru(8, -1, (64, 1, 1), [(4, 64, 64)])
    assert h is not None and not eng.plan_set_tiles(h, 1, 16, 64)      # 8 threads per line < one tile
    eng.plan_destroy(h)
    h = eng.plan_create_guru(4, -1, (256, 1, 1), [(4, 256, 256)])
    assert h is not None and not eng.plan_set_tiles(h, 1, 32, 128) and eng.plan_set_tiles(h, 1, 16, 64)
    eng.plan_destroy(h)


@pytest.mark.parametrize('dt,tile,n,W', [('D', 16, 256, 48), ('F', 32, 2048, 64), ('F', 32, 256, 96), ('D', 16, 1024, 32)])
@pytest.mark.parametrize('side', [0, 1])
def test_strided_plans_address_tile_major_columns(dt, tile, n, W, side):
    import torch
    from mpi4py_fft_amd import _lib
    eng = _lib.engine()
    prec = _lib.precision_of(dt)
    n0 = 3
    rng = np.random.default_rng(6)
    x = (rng.standard_normal((n0, n, W)) + 1j * rng.standard_normal((n0, n, W))).astype(dt)
    want = _natural_fft(x, 1, prec)
    # tile-major slab: [i0][W / tile][n][tile]
    til = dict(es=tile, o=n * W, tS=n * tile)
    nat = dict(es=W, o=n * W)
    si, so = (til, nat) if side == 0 else (nat, til)
    h = eng.plan_create_guru(prec, -1, (n, si['es'], so['es']), [(n0, si['o'], so['o']), (W, 1, 1)])
    assert h is not None and eng.plan_set_tiles(h, side, tile, n * tile)
    xin = x if side == 1 else x.reshape(n0, n, W // tile, tile).transpose(0, 2, 1, 3)
    a = _dev(xin)
    b = torch.zeros(n0 * n * W, dtype=a.dtype, device='cuda')
    eng.execute_ptr(h, a.data_ptr(), b.data_ptr(), 1.0)
    torch.cuda.synchronize()
    got = _host(b)
    ref = want if side == 0 else want.reshape(n0, n, W // tile, tile).transpose(0, 2, 1, 3)
    assert np.array_equal(got.reshape(ref.shape), ref)
    eng.plan_destroy(h)


@pytest.mark.parametrize('dt,W,n', [('F', 33, 256), ('F', 513, 64), ('D', 17, 128), ('D', 9, 1024)])
def test_flat_tiles_read_rows_stored_as_body_plus_leftover_columns(dt, W, n):
    """The last stage of a forward r2c transform on a rank whose half-spectrum block is odd-wide:
    input slabs [rows][body] + [rows][leftover], output the caller's natural (n, rows, W) array."""
    import torch
    from mpi4py_fft_amd import _lib
    eng = _lib.engine()
    prec = _lib.precision_of(dt)
    lw = 128 // np.dtype(dt).itemsize
    bw, n1 = W - W % lw, 6
    tw = W - bw
    rng = np.random.default_rng(7)
    x = (rng.standard_normal((n, n1, W)) + 1j * rng.standard_normal((n, n1, W))).astype(dt)
    want = _natural_fft(x, 0, prec)
    E = n1 * W + 5                                    # slab pitch with some skew
    buf = np.zeros((n, E), dtype=dt)
    buf[:, :n1 * bw] = x[:, :, :bw].reshape(n, n1 * bw)
    buf[:, n1 * bw:n1 * W] = x[:, :, bw:].reshape(n, n1 * tw)
    # (three batch dims: the flattened pair is the last two)
    h = eng.plan_create_guru(prec, -1, (n, E, n1 * W), [(1, 0, 0), (n1, bw, W), (W, 1, 1)])
    assert h is not None and eng.plan_set_flat(h, bw, n1 * bw, tw)
    a = _dev(buf)
    b = torch.zeros(n * n1 * W, dtype=a.dtype, device='cuda')
    eng.execute_ptr(h, a.data_ptr(), b.data_ptr(), 1.0)
    torch.cuda.synchronize()
    assert np.array_equal(_host(b).reshape(n, n1, W), want)
    eng.plan_destroy(h)


@pytest.mark.parametrize('dt,n,p,tile', [('f', 2048, 2, 32), ('f', 512, 4, 32), ('d', 1024, 2, 16), ('d', 256, 3, 16), ('f', 64, 2, 32)])
def test_packed_real_rows_write_and_read_slab_wise_uneven_blocks(dt, n, p, tile):
    import torch
    from mpi4py_fft_amd import _lib
    from mpi4py_fft_amd.pencil import _blockdist
    eng = _lib.engine()
    prec = _lib.precision_of(dt)
    n0, n1, nh = 3, 4, n // 2 + 1
    rng = np.random.default_rng(8)
    x = rng.standard_normal((n0, n1, n)).astype(dt)
    cdt = np.dtype(dt.upper())
    # reference: the plain uneven split ([block][row][w_b]) of the same plan
    rows = n0 * n1
    h0 = eng.plan_create((rows, n), (rows, nh), (1,), _lib.R2C, prec)
    assert eng.plan_set_split(h0, 1, p)
    a = _dev(x)
    b0 = torch.zeros(rows * nh, dtype=torch.from_numpy(np.zeros(1, cdt)).dtype, device='cuda')
    eng.execute_ptr(h0, a.data_ptr(), b0.data_ptr(), 1.0)
    h1 = eng.plan_create((rows, n), (rows, nh), (1,), _lib.R2C, prec)
    assert eng.plan_set_split_slabs(h1, 1, p, n1, tile)
    b1 = torch.zeros_like(b0)
    eng.execute_ptr(h1, a.data_ptr(), b1.data_ptr(), 1.0)
    torch.cuda.synchronize()
    g0, g1 = _host(b0), _host(b1)
    pos = 0
    for r in range(p):
        w, _ = _blockdist(nh, p, r)
        blk = g0[pos:pos + rows * w].reshape(n0, n1, w)
        bw = w - w % tile
        want = np.concatenate([blk[:, :, :bw].reshape(n0, -1), blk[:, :, bw:].reshape(n0, -1)], axis=1)
        assert np.array_equal(g1[pos:pos + rows * w].reshape(n0, n1 * w), want), (r, w)
        pos += rows * w
    # c2r reads the same buffer back: x * n
    h2 = eng.plan_create((rows, nh), (rows, n), (1,), _lib.C2R, prec)
    assert eng.plan_set_split_slabs(h2, 0, p, n1, tile)
    c = torch.zeros_like(a)
    eng.execute_ptr(h2, b1.data_ptr(), c.data_ptr(), 1.0 / n)
    torch.cuda.synchronize()
    assert np.abs(_host(c) - x).max() <= (1e-12 if dt == 'd' else 2e-5) * np.abs(x).max()
    for h in (h0, h1, h2):
        eng.plan_destroy(h)


def test_slab_wise_split_can_be_repeated_and_switched_off():
    """gfft_plan_set_split_slabs starts from the plan's natural row strides on every call (the slab form multiplies
    them) and one block switches the layout off: the plan is then the natural one again, as the header promises."""
    import torch
    from mpi4py_fft_amd import _lib
    eng = _lib.engine()
    n, n0, n1, p, tile = 256, 3, 4, 3, 16
    nh, rows = n // 2 + 1, 12
    x = np.random.default_rng(18).standard_normal((n0, n1, n))
    a = _dev(x)

    def run(h):
        b = torch.zeros(rows * nh, dtype=torch.complex128, device='cuda')
        eng.execute_ptr(h, a.data_ptr(), b.data_ptr(), 1.0)
        torch.cuda.synchronize()
        return _host(b).copy()
    once = eng.plan_create((rows, n), (rows, nh), (1,), _lib.R2C, 8)
    assert eng.plan_set_split_slabs(once, 1, p, n1, tile)
    twice = eng.plan_create((rows, n), (rows, nh), (1,), _lib.R2C, 8)
    assert eng.plan_set_split_slabs(twice, 1, p, n1, tile) and eng.plan_set_split_slabs(twice, 1, p, n1, tile)
    assert np.array_equal(run(once), run(twice))
    with pytest.raises(RuntimeError):
        eng.plan_set_split_slabs(twice, 1, p, 5, tile)              # 5 does not divide the 12 rows: refused ...
    assert np.array_equal(run(once), run(twice))                   # ... and the plan is as it was
    assert eng.plan_set_split_slabs(twice, 1, 1, n1, tile)          # one block: the natural layout again
    want = np.fft.rfft(x, axis=2).reshape(rows, nh)
    got = run(twice).reshape(rows, nh)
    assert np.abs(got - want).max() <= 1e-12 * np.abs(want).max()
    for h in (once, twice):
        eng.plan_destroy(h)


def test_tile_major_rows_on_top_of_a_split_set_after_planning():
    """gfft_plan_set_split followed by gfft_plan_set_tiles on a row plan: the block jump is re-derived from the blocks
    the split recorded (it used to be computed from the guru plan's fields only, i.e. reset to 0)."""
    import torch
    from mpi4py_fft_amd import _lib
    eng = _lib.engine()
    n, rows, nb, tile = 1024, 6, 2, 16
    x = (np.random.default_rng(19).standard_normal((rows, n)) + 1j * np.random.default_rng(20).standard_normal((rows, n)))
    want = np.fft.fft(x, axis=1)
    a = _dev(x)
    h = eng.plan_create((rows, n), (rows, n), (1,), _lib.C2C_FORWARD, 8)
    assert eng.plan_set_split(h, 1, nb)
    per = n // nb
    # blocks of `per` entries per row, block b of all rows first; inside a block the line is tile-major with the
    # tiles of one row back to back (tile stride = tile): the same bytes as the plain split layout
    assert eng.plan_set_tiles(h, 1, tile, tile)
    b = torch.zeros(rows * n, dtype=torch.complex128, device='cuda')
    eng.execute_ptr(h, a.data_ptr(), b.data_ptr(), 1.0)
    torch.cuda.synchronize()
    got = _host(b).reshape(nb, rows, per)
    for blk in range(nb):
        assert np.abs(got[blk] - want[:, blk * per:(blk + 1) * per]).max() <= 1e-12 * np.abs(want).max(), blk
    eng.plan_destroy(h)
