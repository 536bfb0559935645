"""Shared test bodies: the same PFFT / Transfer checks run against the HIP engine on a GPU
(-m gpu) and against the host checker engine on CPU (host-logic tests)."""
import json
import os

import numpy as np

from oracle import pfft_oracle as O
from tests import thread_comm

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load(name):
    return np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False)


def pfft_case_names():
    p = load('pfft')
    return sorted({k.split('/')[0] for k in p.files})


EPS = {'f': 2.0 ** -23, 'F': 2.0 ** -23, 'd': 2.0 ** -52, 'D': 2.0 ** -52}
CONTRACT = {'fwd': {'f': 2e-4, 'F': 2e-4, 'd': 2e-10, 'D': 2e-10}, 'rt': {'f': 1e-4, 'F': 1e-4, 'd': 1e-10, 'D': 1e-10}}


def rounding_tol(dt, nelem, factor=2.0):
    """What a transform of `nelem` points may lose to ROUNDING: factor * eps * log2(nelem) -- 1.3e-14 in fp64, 7e-6 in fp32 at
    1024^3.  Measured over the whole GPU suite (GFFT_TEST_RATIO_LOG, 19 371 guarded comparisons, profiles/r06_guard_ratios.txt):
    the worst case sits at 0.71 eps log2(nelem) -- 7-point lines --, the BASELINE sizes at 0.1 ... 0.4.  The contract tolerances
    (north_star: 1e-10 round trip / 2e-10 forward in fp64; 1e-4 / 2e-4 in fp32) are 10^3 ... 10^5 times looser: a kernel
    that lost three digits would pass them.  The reference's own serial tests ask for the same level
    (/root/reference/tests/test_libfft.py:17, abstol f = 5e-5, d = 1e-14 on O(1) data)."""
    return factor * EPS[dt] * max(1.0, float(np.log2(max(2.0, float(nelem)))))


_RATIO_LOG = os.environ.get('GFFT_TEST_RATIO_LOG')


def _note(kind, dt, nelem, err):
    # calibration aid: GFFT_TEST_RATIO_LOG=<file> records every guarded error against its rounding allowance
    if _RATIO_LOG:
        with open(_RATIO_LOG, 'a') as f:
            f.write('%s %s %d %.3e %.3f\n' % (kind, dt, nelem, err, err / rounding_tol(dt, nelem, 1.0)))


def tol_for(dt, nelem=None):
    """forward vs oracle: max|d| <= tol * max|ref|.  Without `nelem` the contract tolerance alone;
    with the number of transformed points the rounding-level guard as well (whichever is tighter)."""
    if nelem is None:
        return CONTRACT['fwd'][dt]
    return min(CONTRACT['fwd'][dt], rounding_tol(dt, nelem))


def assert_close(got, ref, dt, nelem, what=''):
    """max|got - ref| / max|ref| within the contract tolerance AND at rounding level; returns the error."""
    err = float(np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-30))
    _note('fwd', dt, int(nelem), err)
    assert err <= CONTRACT['fwd'][dt], (what, 'contract', err)
    assert err <= rounding_tol(dt, nelem), (what, 'ROUNDING-LEVEL guard: %.3e > %.3e' % (err, rounding_tol(dt, nelem)))
    return err


def assert_roundtrip(back, orig, dt, nelem, what=''):
    """||back - orig||_2 / ||orig||_2 within the contract tolerance AND at rounding level; returns the error."""
    rel = float(np.linalg.norm(back - orig) / max(np.linalg.norm(orig), 1e-30))
    _note('rt', dt, int(nelem), rel)
    assert rel <= CONTRACT['rt'][dt], (what, 'contract', rel)
    assert rel <= rounding_tol(dt, nelem), (what, 'ROUNDING-LEVEL guard: %.3e > %.3e' % (rel, rounding_tol(dt, nelem)))
    return rel


def run_ranks(P, fn):
    from mpi4py_fft_amd import comm
    if P == 1:
        return [fn(comm.COMM_SELF)]
    return thread_comm.run(P, fn)


def check_pfft_golden(name):
    """PFFT of the product package on P (thread-)ranks vs the fixture the reference produced."""
    from mpi4py_fft_amd import PFFT, newDistArray
    p = load('pfft')
    P = int(p[name + '/P'])
    dt = str(p[name + '/dtype'])
    shape = tuple(int(s) for s in p[name + '/shape'])
    kw = json.loads(str(p[name + '/kw']))
    if 'axes' in kw:
        kw['axes'] = [tuple(a) if isinstance(a, list) else a for a in kw['axes']]
    gin = p[name + '/input']

    def body(comm):
        k = {a: (list(b) if isinstance(b, list) else b) for a, b in kw.items()}
        fft = PFFT(comm, shape, dtype=dt, **k)
        u = newDistArray(fft, False)
        u[...] = gin[fft.local_slice(False)]
        uh = fft.forward(u)
        uh_host = np.asarray(uh).copy()
        ub = np.asarray(fft.backward(uh)).copy()
        pin, pout = fft.pencil
        info = dict(axes=[list(a) for a in fft.axes], grid=[c.Get_size() for c in fft.subcomm],
                    in_subshape=list(pin.subshape), in_substart=list(pin.substart), in_axis=pin.axis,
                    out_subshape=list(pout.subshape), out_substart=list(pout.substart),
                    out_axis=pout.axis, gshape_in=list(fft.global_shape(False)),
                    gshape_out=list(fft.global_shape(True)), nxfftn=len(fft.xfftn),
                    ntransfer=len(fft.transfer),
                    transfers=[[t.comm.Get_size(), list(t.shape), list(t.subshapeA), t.axisA,
                                list(t.subshapeB), t.axisB] for t in fft.transfer])
        # structural asserts of tests/test_mpifft.py:144-164
        assert fft.dtype(True) == fft.forward.output_array.dtype
        assert fft.dtype(False) == fft.forward.input_array.dtype
        assert len(fft.axes) == len(fft.xfftn) == len(fft.transfer) + 1
        assert fft.forward.input_pencil.subshape == fft.forward.input_array.shape
        assert fft.forward.output_pencil.subshape == fft.forward.output_array.shape
        assert fft.backward.input_pencil.subshape == fft.backward.input_array.shape
        assert fft.backward.output_pencil.subshape == fft.backward.output_array.shape
        assert fft.dimensions == len(shape)
        fft.destroy()
        return info, uh_host, ub

    res = run_ranks(P, body)
    tol = tol_for(dt)
    nelem = int(np.prod(shape))
    for r, (info, uh, ub) in enumerate(res):
        ref_info = json.loads(str(p['%s/r%d/info' % (name, r)]))
        for key in ref_info:
            assert info[key] == ref_info[key], (name, r, key, info[key], ref_info[key])
        ref = p['%s/r%d/fwd' % (name, r)]
        assert uh.shape == ref.shape and uh.dtype == ref.dtype, (name, uh.shape, ref.shape)
        assert np.abs(uh - ref).max() <= tol * max(np.abs(ref).max(), 1e-30), \
            (name, r, 'forward', np.abs(uh - ref).max())
        assert_close(uh, ref, dt, nelem, (name, r, 'forward'))            # ... and at rounding level
        refb = p['%s/r%d/bwd' % (name, r)]
        assert ub.shape == refb.shape and ub.dtype == refb.dtype
        assert np.abs(ub - refb).max() <= 10 * tol * max(np.abs(refb).max(), 1e-30), \
            (name, r, 'backward', np.abs(ub - refb).max())
        assert np.abs(ub - refb).max() <= 2 * rounding_tol(dt, nelem) * max(np.abs(refb).max(), 1e-30), (name, r, 'backward, rounding level')


def check_transfer_golden(ci):
    """tests/test_pencil.py's chain on the product Transfer vs what the reference's Alltoallw
    delivered (fixture), plus the fwd.fwd.bwd.bwd identity."""
    from mpi4py_fft_amd import Subcomm, Pencil, asdevice, zeros
    t = load('transfer')
    key = 'transfer%d' % ci
    P = int(t[key + '/P'])
    shape = tuple(int(s) for s in t[key + '/shape'])
    a1, a2, a3 = (int(a) for a in t[key + '/axes'])
    pdim = int(t[key + '/pdim'])
    pdim = None if pdim < 0 else pdim
    G = np.arange(int(np.prod(shape)), dtype='d').reshape(shape)

    def body(comm):
        subcomm = Subcomm(comm, pdim)
        p0 = Pencil(subcomm, shape)
        pA = p0.pencil(a1)
        pB = pA.pencil(a2)
        pC = pB.pencil(a3)
        t1 = Pencil.transfer(pA, pB, 'd')
        t2 = Pencil.transfer(pB, pC, 'd')
        sl = tuple(slice(s, s + n) for s, n in zip(pA.substart, pA.subshape))
        A = asdevice(np.ascontiguousarray(G[sl]))
        B = zeros(pB.subshape)
        C = zeros(pC.subshape)
        t1.forward(A, B)
        t2.forward(B, C)
        A2, B2 = zeros(pA.subshape), zeros(pB.subshape)
        t2.backward(C, B2)
        t1.backward(B2, A2)
        geo = np.array([pA.subshape, pA.substart, pB.subshape, pB.substart, pC.subshape, pC.substart])
        return geo, np.asarray(A), np.asarray(B), np.asarray(C), np.asarray(A2)

    for r, (geo, A, B, C, A2) in enumerate(run_ranks(P, body)):
        assert np.array_equal(geo, t['%s/r%d/geo' % (key, r)])
        assert np.array_equal(A, t['%s/r%d/A' % (key, r)])
        assert np.array_equal(B, t['%s/r%d/B' % (key, r)]), (key, r)
        assert np.array_equal(C, t['%s/r%d/C' % (key, r)]), (key, r)
        assert np.array_equal(A2, A)


def check_pfft_vs_oracle(P, shape, dt, seed=7, **kw):
    """Product PFFT on P ranks vs the oracle on the same seeded input: forward values,
    round trip, and the global-DFT identity for unpadded transforms."""
    from mpi4py_fft_amd import PFFT, newDistArray
    offt = O.OPFFT(P, shape, dtype=dt, **{k: (list(v) if isinstance(v, list) else v) for k, v in kw.items()})
    G = O.rng_array(offt.input_shape, dt, seed)
    ref = offt.forward(offt.scatter(G))

    def body(comm):
        k = {a: (list(b) if isinstance(b, list) else b) for a, b in kw.items()}
        if 'r2r' in k:
            # the oracle's {axes: FFTW kind} spelled as the product's `transforms=` dict of planners
            import functools
            from mpi4py_fft_amd import fftw
            tr = {}
            for axes_, kind in k.pop('r2r').items():
                fam, typ = O.R2R_KINDS[kind]
                fwd, bck = (fftw.dctn, fftw.idctn) if fam == 'dct' else (fftw.dstn, fftw.idstn)
                tr[tuple(axes_)] = (functools.partial(fwd, type=typ), functools.partial(bck, type=typ))
            k['transforms'] = tr
        fft = PFFT(comm, shape, dtype=dt, **k)
        u = newDistArray(fft, False)
        u[...] = G[fft.local_slice(False)]
        uh = np.asarray(fft.forward(u)).copy()
        back = np.asarray(fft.backward()).copy()
        sl = fft.local_slice(False)
        fft.destroy()
        return uh, back, sl

    tol = tol_for(dt)
    for r, (uh, back, sl) in enumerate(run_ranks(P, body)):
        assert uh.shape == ref[r].shape and uh.dtype == ref[r].dtype
        scale = max(np.abs(ref[r]).max(), 1e-30)
        assert np.abs(uh - ref[r]).max() <= tol * scale, (shape, dt, kw, np.abs(uh - ref[r]).max() / scale)
        nelem = int(np.prod(offt.input_shape))
        assert_close(uh, ref[r], dt, nelem, (shape, dt, kw))              # ... and at rounding level
        if not kw.get('padding'):
            d = back - G[sl]
            rel = np.linalg.norm(d) / np.linalg.norm(G[sl])
            assert rel <= (1e-10 if dt in 'dD' else 1e-4), (shape, dt, kw, rel)
            assert_roundtrip(back, G[sl], dt, nelem, (shape, dt, kw))


def check_redistribute_chain(rng, mid=False):
    """A random DistArray (shape, tensor rank, dtype, grid, alignment) walked through four random
    redistributions on thread ranks; after each one every rank must hold exactly its local_slice() of
    the global array.  Returns False when the draw is not a valid configuration."""
    from mpi4py_fft_amd import DistArray
    nd = int(rng.choice([2, 3, 3, 4]))
    pool = [5, 8, 9, 12, 13, 16, 17, 24, 31, 32, 33, 40, 64] if not mid else [16, 33, 64, 100, 128, 129, 256, 257, 512, 513]
    P = int(rng.choice([2, 3, 4, 4, 6, 8, 8]))
    shape = tuple(int(rng.choice(pool)) for _ in range(nd))
    if min(shape) < P or np.prod(shape) > (24_000_000 if mid else 2_000_000):
        return False
    rank = int(rng.choice([0, 0, 1, 2])) if not mid else int(rng.choice([0, 0, 1]))
    dt = str(rng.choice(list('fdFD')))
    align = int(rng.integers(0, nd))
    # how many axes are distributed: 1 (slab) ... nd - 1
    ndist = int(rng.integers(1, nd))
    walk = [int(a) for a in rng.integers(0, nd, size=4)]
    comps = (3,) * rank
    G = (rng.standard_normal(comps + shape) + (1j * rng.standard_normal(comps + shape) if dt in 'FD' else 0)).astype(dt)

    def body(comm):
        from mpi4py_fft_amd.pencil import Subcomm
        dims = [0] * nd
        # the aligned axis is undivided; distribute `ndist` of the others
        others = [a for a in range(nd) if a != align]
        for a in others[ndist:]:
            dims[a] = 1
        dims[align] = 1
        sub = Subcomm(comm, dims)
        a = DistArray(comps + shape, subcomm=sub, dtype=dt, alignment=align, rank=rank)
        sl = a.local_slice()
        a[...] = G[sl]
        bad = []
        cur = a
        for ax in walk:
            cur = cur.redistribute(ax)
            got = np.asarray(cur)
            if cur.alignment != ax and cur.commsizes[rank + ax] != 1:
                bad.append(('alignment', ax, cur.alignment))
            if not np.array_equal(got, G[cur.local_slice()]):
                bad.append(('values', ax, got.shape))
        return bad
    res = run_ranks(P, body)
    assert not any(res), (P, shape, dt, rank, align, ndist, walk, [r for r in res if r][:1])
    return True


def impulse_errors(n, dt, strided, positions=None, lines=16):
    """Twiddle integrity of ONE serial plan of length n: line b of a batch holds a unit impulse at position j_b, so its
    transform is X_b[k] = exp(-2 pi i j_b k / n) -- every output a bare product of the plan's twiddle factors, of modulus
    1: a wrong table entry shows at full size in the outputs it feeds (random data dilutes it by ~1 / (3.5 sqrt(radix)),
    below fp32 rounding for an entry off by 1e-5).  Returns (max |forward - exact|, max |backward(exact) - impulse|),
    in units of 1 (no normalisation needed: |X| = 1, impulse height 1)."""
    from mpi4py_fft_amd import FFT, asdevice
    cdt = 'D' if dt in 'dD' else 'F'
    if positions is None:
        positions = [1, 2, 3, 5, 7, n // 2 + 1, n - 1, n // 3 + 1, n // 4 + 1, 11 % n, n // 5 + 2, n // 7 + 3, 1, n - 2, 13 % n, n // 2 - 1]
    positions = [int(j) % n for j in positions][:lines]
    B = len(positions)
    shape, axis = ((n, B), 0) if strided else ((B, n), 1)
    x = np.zeros(shape, dtype=cdt)
    k = np.arange(n)
    want = np.empty(shape, dtype='D')
    for b, j in enumerate(positions):
        col = np.exp(-2j * np.pi * ((j * k) % n) / n)
        if strided:
            x[j, b] = 1
            want[:, b] = col
        else:
            x[b, j] = 1
            want[b] = col
    fft = FFT(shape, (axis,), dtype=cdt)
    got = np.asarray(fft.forward(asdevice(x), normalize=False)).astype('D')
    ef = float(np.abs(got - want).max())
    back = np.asarray(fft.backward(asdevice(want.astype(cdt)), normalize=True)).astype('D')
    eb = float(np.abs(back - x).max())
    fft.destroy()
    return ef, eb
