"""Pins oracle/pfft_oracle.py against fixtures produced by the reference's own Python
(oracle/make_golden.py) and the reference's docstring known-answer vectors."""
import json

import numpy as np
import pytest

from oracle import pfft_oracle as O


def test_blockdist(golden):
    for N, p, r, n, s in golden['blockdist']['blockdist']:
        assert O.blockdist(N, p, r) == (n, s)


def test_compute_dims_matches_appendix_a():
    assert O.compute_dims(1, [0, 0, 1]) == [1, 1, 1]
    assert O.compute_dims(2, [0, 0, 1]) == [2, 1, 1]
    assert O.compute_dims(4, [0, 0, 1]) == [2, 2, 1]
    assert O.compute_dims(8, [0, 0, 1]) == [4, 2, 1]
    assert O.compute_dims(6, [0, 0, 1]) == [3, 2, 1]
    assert O.compute_dims(8, [0, 1, 1]) == [8, 1, 1]
    assert O.compute_dims(4, [1, 0]) == [1, 4]


def test_baseline_geometry(golden):
    g = golden['geometry']
    for key in g.files:
        name, P = key.split('_P')
        P = int(P)
        shape = {'C3': (512,) * 3, 'C4': (1024,) * 3, 'C5': (2048,) * 3}[name]
        real = name == 'C5'
        dims = O.compute_dims(P, [0, 0, 1])
        for r in range(P):
            ref = g[key][r]
            assert list(ref[8]) == dims
            c = O.rank_coords(r, dims)
            p0 = O.OPencil(dims, c, shape, 2)
            shp = list(shape)
            if real:
                shp[2] = shp[2] // 2 + 1
            pA = O.OPencil(dims, c, shp, 2)
            pB = pA.pencil(1)
            pC = pB.pencil(0)
            got = [p0.subshape, p0.substart, pA.subshape, pA.substart, pB.subshape,
                   pB.substart, pC.subshape, pC.substart]
            assert np.array_equal(np.array(got), ref[:8])


def test_kats(golden):
    k = golden['libfft']
    f = O.OFFT((4,), None, 'D')
    assert np.allclose(f.forward(k['kat/fftn_in'], normalize=False), k['kat/fftn_out'], atol=1e-14)
    f = O.OFFT((4,), None, 'd')
    assert np.allclose(f.forward(k['kat/rfftn_in'], normalize=False), k['kat/rfftn_out'], atol=1e-14)
    f = O.OFFT((6,), None, 'd')
    assert np.allclose(f.backward(k['kat/irfftn_in']), k['kat/irfftn_out6'], atol=1e-13)
    f = O.OFFT((7,), None, 'd')
    assert np.allclose(f.backward(k['kat/irfftn_in']), k['kat/irfftn_out7'], atol=1e-7)


def test_libfft_cases(golden):
    k = golden['libfft']
    i = 0
    while 'libfft%d/A' % i in k.files:
        p = 'libfft%d/' % i
        dt = str(k[p + 'dtype'])
        axes = [int(a) for a in k[p + 'axes']]
        axes = None if axes == [-99] else axes
        pad = float(k[p + 'padding'])
        f = O.OFFT(tuple(k[p + 'shape']), axes, dt, padding=(pad if pad else False))
        tol = 1e-14 if dt in 'dD' else 2e-6
        B = f.forward(k[p + 'A'])
        assert B.shape == k[p + 'B'].shape and B.dtype == k[p + 'B'].dtype
        assert np.abs(B - k[p + 'B']).max() <= tol * max(1, np.abs(k[p + 'B']).max())
        A2 = f.backward(k[p + 'B'])
        assert A2.shape == k[p + 'A2'].shape and A2.dtype == k[p + 'A2'].dtype
        assert np.abs(A2 - k[p + 'A2']).max() <= 20 * tol * max(1, np.abs(k[p + 'A2']).max())
        i += 1
    assert i >= 18


def test_transfer_block_maps(golden):
    t = golden['transfer']
    ci = 0
    while 'transfer%d/P' % ci in t.files:
        key = 'transfer%d' % ci
        P = int(t[key + '/P'])
        shape = tuple(t[key + '/shape'])
        a1, a2, a3 = (int(a) for a in t[key + '/axes'])
        pdim = int(t[key + '/pdim'])
        nd = len(shape)
        # Subcomm(comm, pdim): pdim=None -> dims=[0]; int -> [0]*pdim   (pencil.py:68-79)
        dims = O.compute_dims(P, [0] if pdim < 0 else [0] * pdim)
        grid = [int(x) for x in t[key + '/r0/grid']]
        assert dims == grid
        # Pencil(subcomm, shape) with a short subcomm: COMM_SELF appended / inserted at axis
        axis0 = nd - 1
        sizes = list(dims) + [1] * (nd - 1 - len(dims))
        sizes.insert(axis0, 1)
        G = np.arange(int(np.prod(shape)), dtype='d').reshape(shape)
        pens = []
        for r in range(P):
            c = list(O.rank_coords(r, dims)) + [0] * (nd - 1 - len(dims))
            c.insert(axis0, 0)
            pens.append(O.OPencil(sizes, c, shape, axis0).pencil(a1))
        A = [np.ascontiguousarray(G[p.local_slice()]) for p in pens]
        for r in range(P):
            geo = t['%s/r%d/geo' % (key, r)]
            assert np.array_equal(geo[0], pens[r].subshape) and np.array_equal(geo[1], pens[r].substart)
            assert np.array_equal(A[r], t['%s/r%d/A' % (key, r)])
        # the result of a redistribution is the global array re-sliced by the new pencil
        pB = [p.pencil(a2) for p in pens]
        pC = [p.pencil(a3) for p in pB]
        for r in range(P):
            assert np.array_equal(G[pB[r].local_slice()], t['%s/r%d/B' % (key, r)])
            assert np.array_equal(G[pC[r].local_slice()], t['%s/r%d/C' % (key, r)])
        ci += 1
    assert ci == 6


def test_transfer_pack_exchange_unpack(golden):
    """O.transfer (explicit pack/exchange/unpack) reproduces what the reference's Alltoallw moved."""
    t = golden['transfer']
    key = 'transfer5'           # P=8, shape (9,8,16), axes (2,0,1), grid (4,2)
    P = int(t[key + '/P'])
    shape = tuple(t[key + '/shape'])
    dims = [int(x) for x in t[key + '/r0/grid']] + [1]
    A = [t['%s/r%d/A' % (key, r)] for r in range(P)]
    geoB = [t['%s/r%d/geo' % (key, r)][2] for r in range(P)]
    # pencil A is aligned on axis 2 (split over grid axes 0,1 along array axes 0,1); B aligned on 0:
    # the communicator is the one array axis 0 was distributed over = grid axis 0
    out = [None] * P
    for grp in O.transfer_groups(dims, 0):
        res = O.transfer(A, grp, 2, 0, shape[0])
        for g, a in zip(grp, res):
            out[g] = a
    for r in range(P):
        assert tuple(geoB[r]) == out[r].shape
        assert np.array_equal(out[r], t['%s/r%d/B' % (key, r)])


def _cases(p):
    return sorted({k.split('/')[0] for k in p.files})


def test_pfft_cases(golden):
    p = golden['pfft']
    names = _cases(p)
    assert len(names) >= 20
    for name in names:
        P = int(p[name + '/P'])
        dt = str(p[name + '/dtype'])
        shape = tuple(int(s) for s in p[name + '/shape'])
        kw = json.loads(str(p[name + '/kw']))
        if 'axes' in kw:
            kw['axes'] = [tuple(a) if isinstance(a, list) else a for a in kw['axes']]
        fft = O.OPFFT(P, shape, dtype=dt, **kw)
        gin = p[name + '/input']
        tol = 1e-14 if dt in 'dD' else 1e-6
        assert fft.input_shape == gin.shape
        u = fft.scatter(gin)
        uh = fft.forward(u)
        ub = fft.backward(uh)
        for r in range(P):
            info = json.loads(str(p['%s/r%d/info' % (name, r)]))
            assert [list(a) for a in fft.axes] == info['axes'], name
            assert list(fft.dims) == info['grid']
            assert list(fft.pencil_in[r].subshape) == info['in_subshape']
            assert list(fft.pencil_in[r].substart) == info['in_substart']
            assert fft.pencil_in[r].axis == info['in_axis']
            assert list(fft.pencil_out[r].subshape) == info['out_subshape'], name
            assert list(fft.pencil_out[r].substart) == info['out_substart']
            assert fft.pencil_out[r].axis == info['out_axis']
            assert list(fft.output_shape) == info['gshape_out']
            assert len(fft.transfers) == info['ntransfer']
            ref = p['%s/r%d/fwd' % (name, r)]
            assert uh[r].shape == ref.shape and uh[r].dtype == ref.dtype, name
            assert np.abs(uh[r] - ref).max() <= tol * max(1.0, np.abs(ref).max()), name
            refb = p['%s/r%d/bwd' % (name, r)]
            assert ub[r].shape == refb.shape and ub[r].dtype == refb.dtype
            assert np.abs(ub[r] - refb).max() <= 50 * tol * max(1.0, np.abs(refb).max()), name


def test_pfft_equals_global_dft():
    """Unpadded forward == fftn(global)/N re-sliced by the output pencil, any rank count."""
    for P, shape, dt in ((4, (12, 10, 8), 'D'), (8, (16, 16, 10), 'd'), (2, (9, 7), 'D')):
        fft = O.OPFFT(P, shape, dtype=dt)
        G = O.rng_array(shape, dt, 5)
        uh = fft.forward(fft.scatter(G))
        full = (np.fft.rfftn(G, axes=range(len(shape))) if dt == 'd' else np.fft.fftn(G)) / G.size
        assert np.abs(fft.gather(uh) - full).max() < 1e-15 * G.size


# ---- the plain-C restatement (oracle/dft_oracle.c) -------------------------------------------
def _c_oracle():
    import ctypes
    import os
    import subprocess
    here = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'oracle')
    so = os.path.join(here, 'libdft_oracle.so')
    if not os.path.exists(so):
        subprocess.check_call(['make', '-C', here])
    lib = ctypes.CDLL(so)
    i64p = ctypes.POINTER(ctypes.c_int64)
    dp = ctypes.POINTER(ctypes.c_double)
    lib.dft_oracle_xfftn.argtypes = [ctypes.c_int, i64p, dp, i64p, dp, ctypes.c_int,
                                     ctypes.POINTER(ctypes.c_int), ctypes.c_int]

    def run(a, axes, kind, out_shape, out_dtype):
        a = np.ascontiguousarray(a, dtype='D' if a.dtype.kind == 'c' else 'd')
        out = np.zeros(out_shape, dtype=out_dtype)
        si = (ctypes.c_int64 * a.ndim)(*a.shape)
        so_ = (ctypes.c_int64 * a.ndim)(*out_shape)
        ax = (ctypes.c_int * len(axes))(*axes)
        rc = lib.dft_oracle_xfftn(a.ndim, si, a.view('d').ctypes.data_as(dp), so_,
                                  out.view('d').ctypes.data_as(dp), len(axes), ax, kind)
        assert rc == 0
        return out
    return run


def test_c_oracle_kats_and_fixtures(golden):
    run = _c_oracle()
    k = golden['libfft']
    assert np.allclose(run(k['kat/fftn_in'], [0], -1, (4,), 'D'), k['kat/fftn_out'], atol=1e-14)
    assert np.allclose(run(k['kat/rfftn_in'], [0], -2, (3,), 'D'), k['kat/rfftn_out'], atol=1e-14)
    assert np.allclose(run(k['kat/irfftn_in'], [0], 2, (6,), 'd'), k['kat/irfftn_out6'], atol=1e-13)
    assert np.allclose(run(k['kat/irfftn_in'], [0], 2, (7,), 'd'), k['kat/irfftn_out7'], atol=1e-7)
    # unpadded fp64 libfft fixtures: forward (scaled by 1/M in the reference) and backward
    i = 0
    checked = 0
    while 'libfft%d/A' % i in k.files:
        p = 'libfft%d/' % i
        i += 1
        dt = str(k[p + 'dtype'])
        if float(k[p + 'padding']) or dt in 'fF':
            continue
        A, B, A2 = k[p + 'A'], k[p + 'B'], k[p + 'A2']
        axes = [int(a) for a in k[p + 'axes']]
        axes = list(range(A.ndim)) if axes == [-99] else axes
        M = np.prod([A.shape[a] for a in axes])
        fwd = run(A, axes, -2 if dt == 'd' else -1, B.shape, 'D') / M
        assert np.abs(fwd - B).max() < 1e-14 * max(1, np.abs(B).max())
        bwd = run(B, axes, 2 if dt == 'd' else 1, A.shape, dt)
        assert np.abs(bwd - A2).max() < 1e-13 * max(1, np.abs(A2).max())
        checked += 1
    assert checked >= 8
