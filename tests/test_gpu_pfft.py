"""GPU parity of PFFT / Transfer / DistArray through the product path (HIP engine), on one GPU
with thread-ranks standing in for processes (tests/thread_comm.py: the kernels are real, only
the RCCL wire is replaced by device copies)."""
import itertools

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from tests import cases
from oracle import pfft_oracle as O

cases_tol = cases.tol_for


@pytest.mark.parametrize('name', cases.pfft_case_names())
def test_pfft_matches_reference_fixture(name):
    cases.check_pfft_golden(name)


@pytest.mark.parametrize('ci', range(6))
def test_transfer_matches_reference_fixture(ci):
    cases.check_transfer_golden(ci)


@pytest.mark.parametrize('dt', list('fdFD'))
def test_pencil_loop(dt):
    """tests/test_pencil.py: fwd.fwd.bwd.bwd = identity, sizes (7,8,9), 2 and 4 ranks."""
    from mpi4py_fft_amd import Subcomm, Pencil, asdevice, zeros
    from tests import thread_comm
    for P in (2, 4):
        for dim in (2, 3):
            for shape in itertools.product(*([(7, 8, 9)] * dim)):
                axes = list(range(dim))
                for a1, a2, a3 in itertools.product(axes, axes, axes):
                    if a1 == a2 or a2 == a3 or min(shape) < P:
                        continue
                    if (a1 + a2 + a3 + shape[0]) % 3:      # thin the reference's full product
                        continue
                    for pdim in [None] + list(range(1, dim - 1)):
                        def body(comm):
                            sub = Subcomm(comm, pdim)
                            pA = Pencil(sub, shape).pencil(a1)
                            pB = pA.pencil(a2)
                            pC = pB.pencil(a3 - len(shape))
                            t1, t2 = Pencil.transfer(pA, pB, dt), Pencil.transfer(pB, pC, dt)
                            X = np.random.default_rng(comm.Get_rank()).random(pA.subshape).astype(dt)
                            A, B, C = asdevice(X), zeros(pB.subshape, dt), zeros(pC.subshape, dt)
                            t1.forward(A, B)
                            t2.forward(B, C)
                            B.fill(0)
                            t2.backward(C, B)
                            A.fill(0)
                            t1.backward(B, A)
                            return np.array_equal(np.asarray(A), X)
                        assert all(thread_comm.run(P, body)), (P, shape, a1, a2, a3, pdim)


def _allaxes(dim):
    if dim == 2:
        return [None, (-1,), (-2,), (-1, -2), (-2, -1), (-1, 0), (0, -1), ((0,), (1,))]
    if dim == 3:
        return [None, ((0,), (1, 2)), ((0,), (-2, -1))]
    return [None, ((0,), (1,), (2,), (3,)), ((0,), (1, 2, 3)), ((0,), (1,), (2, 3))]


@pytest.mark.parametrize('P', [1, 2, 4])
@pytest.mark.parametrize('dt', list('fdFD'))
def test_mpifft_loop(P, dt):
    """tests/test_mpifft.py:57-177 (sizes 12/13, dims 2-4, grids, collapse, axes spellings),
    checking values against the oracle as well as the round trip."""
    for dim in (2, 3, 4):
        shapes = list(itertools.product(*([(12, 13)] * dim)))
        for shape in shapes[::3] if dim == 4 else shapes:
            if dim < 3:
                n = min(shape)
                if dt in 'fd':
                    n = n // 2 + 1
                if n < P:
                    continue
            for grid in ((None,) if dim == 2 else ((-1,), None)):
                for collapse in (True, False):
                    for axes in _allaxes(dim):
                        g = grid
                        if grid is not None:
                            ax = -1
                            if axes is not None:
                                ax = axes[-1] if isinstance(axes[-1], int) else axes[-1][-1]
                            slab = (ax + 1) % len(shape)
                            g = [1] * (slab + 1)
                            g[slab] = 0
                        kw = dict(collapse=collapse)
                        if axes is not None:
                            kw['axes'] = axes
                        if g is not None:
                            kw['grid'] = g
                        if dim == 4 and dt in 'fd':
                            # real 4-D cases run DCT-III stages wherever the trailing groups match
                            # (tests/test_mpifft.py:97-110)
                            kw['r2r'] = {(3,): 4, (2, 3): 4, (1, 2, 3): 4, (0, 1, 2, 3): 4}
                        cases.check_pfft_vs_oracle(P, shape, dt, **kw)


@pytest.mark.parametrize('P', [1, 2, 4])
@pytest.mark.parametrize('dt', list('dDF'))
def test_mpifft_padding(P, dt):
    """tests/test_mpifft.py:181-251: padding 1.5, fwd.bwd.fwd idempotence and swapped normalize."""
    from mpi4py_fft_amd import PFFT, newDistArray
    from oracle import pfft_oracle as O
    for shape, axes in (((12, 13), None), ((12, 13), (-2, -1)), ((12, 13, 12), None),
                        ((13, 12, 12), ((0,), (1,), (2,))), ((12, 12, 13, 12), ((0,), (1,), (2,), (3,)))):
        if P > 1 and len(shape) == 2 and dt == 'd':
            continue
        padding = [1.5] * len(shape)
        kw = dict(padding=padding)
        if axes is not None:
            kw['axes'] = axes
        cases.check_pfft_vs_oracle(P, shape, dt, **kw)

        def body(comm):
            fft = PFFT(comm, shape, dtype=dt, **{k: (list(v) if isinstance(v, list) else v) for k, v in kw.items()})
            U = np.random.default_rng(comm.Get_rank()).random(fft.forward.input_array.shape).astype(dt)
            F = np.asarray(fft.forward(U)).copy()
            fft.backward.input_array[...] = F
            fft.backward()
            fft.forward()
            ok1 = np.allclose(np.asarray(fft.forward.output_array), F, rtol=0, atol=2e-10 if dt in 'dD' else 1e-4)
            fft.backward.input_array[...] = F
            fft.backward(normalize=True)
            fft.forward(normalize=False)
            ok2 = np.allclose(np.asarray(fft.forward.output_array), F, rtol=0, atol=2e-10 if dt in 'dD' else 1e-4)
            fft.destroy()
            return ok1 and ok2
        assert all(cases.run_ranks(P, body)), (shape, axes, dt, P)


def test_distarray_redistribute_and_pfft_darray():
    """tests/test_darray.py: norm conservation under redistribute, rank-1 fields, PFFT(darray=)."""
    from mpi4py_fft_amd import DistArray, newDistArray, PFFT, Subcomm
    from tests import thread_comm

    def body(c):
        N = (8, 10, 12)
        sub = Subcomm(c, [0, 0, 1])
        z = DistArray(N, subcomm=sub, dtype=float, alignment=2)
        z[...] = np.random.default_rng(c.Get_rank()).random(z.shape)
        n0 = sum(c.allgather_obj(float(np.sum(np.asarray(z) ** 2))))
        z1 = z.redistribute(1)
        z0 = z1.redistribute(0)
        n1 = sum(c.allgather_obj(float(np.sum(np.asarray(z1) ** 2))))
        n2 = sum(c.allgather_obj(float(np.sum(np.asarray(z0) ** 2))))
        assert np.isclose(n0, n1) and np.isclose(n0, n2)
        v = DistArray((3,) + N, subcomm=sub, dtype='D', alignment=2, rank=1)
        v[...] = np.random.default_rng(5).random(v.shape) + 0j
        w = v.redistribute(0)
        m0 = sum(c.allgather_obj(float(np.sum(np.abs(np.asarray(v)) ** 2))))
        m1 = sum(c.allgather_obj(float(np.sum(np.abs(np.asarray(w)) ** 2))))
        assert np.isclose(m0, m1)
        fft = PFFT(c, darray=z1)          # aligned axis 1 must be transformed first
        u = newDistArray(fft, False)
        assert u.shape == z1.shape
        u[...] = z1
        uh = fft.forward(u)
        back = fft.backward(uh)
        assert np.allclose(np.asarray(back), np.asarray(z1), atol=1e-12)
        return True

    for P in (1, 2, 4):
        assert all(cases.run_ranks(P, body) if P > 1 else [body(__import__('mpi4py_fft_amd').comm.COMM_SELF)])


def test_nccl_world_of_one():
    """The torch.distributed/RCCL plumbing of comm.TorchComm on the one GPU a test box has."""
    import os
    import torch
    import torch.distributed as dist
    from mpi4py_fft_amd import comm, PFFT, newDistArray
    if dist.is_initialized():
        pytest.skip('process group already initialised')
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29631')
    dist.init_process_group('nccl', rank=0, world_size=1)
    try:
        w = comm.world()
        assert w.Get_size() == 1
        fft = PFFT(w, (16, 12, 10), dtype='D')
        u = newDistArray(fft, False)
        G = np.random.default_rng(0).random(u.shape) + 0j
        u[...] = G
        uh = np.asarray(fft.forward(u))
        assert np.allclose(uh, np.fft.fftn(G) / G.size, atol=1e-14)
        t = torch.arange(8, dtype=torch.float64, device='cuda')
        r = torch.empty_like(t)
        dist.all_to_all_single(r, t, [8], [8])       # RCCL alltoallv code path
        assert torch.equal(r, t)
    finally:
        dist.destroy_process_group()


def test_chunked_transfer_pipeline_on_device(monkeypatch):
    """Slab-chunked asynchronous exchange with the real pack/unpack kernels."""
    from mpi4py_fft_amd import pencil
    monkeypatch.setattr(pencil.Transfer, 'CHUNK_MIN_BYTES', 0)
    monkeypatch.setattr(pencil.Transfer, 'CHUNKS', 3)
    for ci in range(6):
        cases.check_transfer_golden(ci)
    for name in ('c2c_16x16x16_p8', 'r2c_16x16x18_p8', 'r2c_13x12x10_p4', 'c2c_6x7x8x9_p4'):
        cases.check_pfft_golden(name)
    cases.check_pfft_vs_oracle(8, (32, 48, 40), 'D')
    cases.check_pfft_vs_oracle(4, (33, 20, 18), 'd')


def test_chunk_decision_is_the_same_on_every_rank_of_the_group(monkeypatch):
    """An r2c half spectrum of 9 entries over 2 ranks: 5 | 4 columns, so the two ranks' local arrays
    differ in size; a threshold between the two sizes must not make one rank cut its exchange into
    slabs and its peer not (found by tools/stress.py mid: all-to-all counts off by the slab count)."""
    from mpi4py_fft_amd import pencil
    monkeypatch.setenv('GFFT_FUSE_PACK', '0')
    monkeypatch.setenv('GFFT_WIRE', 'torch')
    monkeypatch.setenv('GFFT_RELAY', '0')
    # local arrays at P = 4, grid (2,2): A (8,16,9) 18432 B; B (8,32,5) 20480 B or (8,32,4) 16384 B
    for threshold in (17000, 19000, 21000):
        monkeypatch.setattr(pencil.Transfer, 'CHUNK_MIN_BYTES', threshold)
        cases.check_pfft_vs_oracle(4, (16, 32, 16), 'd')


def test_relayed_exchange_on_device(monkeypatch):
    """Two-round multi-path exchange (relay.py) around the real pack/unpack kernels."""
    monkeypatch.setenv('GFFT_RELAY', '1')
    from mpi4py_fft_amd import relay
    calls, run = [], relay.Schedule.run
    monkeypatch.setattr(relay.Schedule, 'run', lambda self, *a: (calls.append(1), run(self, *a))[1])
    for name in ('c2c_16x16x16_p8', 'r2c_16x16x18_p8', 'r2c_13x12x10_p4', 'c2c_6x7x8x9_p4'):
        cases.check_pfft_golden(name)
    cases.check_pfft_vs_oracle(8, (32, 48, 40), 'D')
    cases.check_pfft_vs_oracle(8, (64, 64, 64), 'F')
    cases.check_pfft_vs_oracle(4, (33, 20, 18), 'd')
    cases.check_pfft_vs_oracle(6, (24, 20, 18), 'd', padding=[1.5, 1.5, 1.5])
    assert calls


def test_pack_fusion_on_device(monkeypatch):
    """Exchange buffers written / read by the FFT kernels themselves (PFFT._fuse_packs): same
    results as the staged pack -> exchange -> unpack path, with and without the relayed route."""
    from tests import thread_comm
    from mpi4py_fft_amd import PFFT

    def flags(comm):
        return [(t.packedA, t.packedB) for t in PFFT(comm, (64, 64, 64), dtype='D').transfer]
    assert thread_comm.run(8, flags)[0] == [(True, True), (True, True)]
    for mode in ('0', '1'):
        monkeypatch.setenv('GFFT_RELAY', mode)
        cases.check_pfft_golden('c2c_16x16x16_p8')
        cases.check_pfft_vs_oracle(8, (64, 64, 64), 'D')
        cases.check_pfft_vs_oracle(8, (64, 128, 32), 'F')
        cases.check_pfft_vs_oracle(4, (128, 96, 64), 'D')
        cases.check_pfft_vs_oracle(4, (64, 64, 64), 'd')
        cases.check_pfft_vs_oracle(2, (64, 64, 64), 'D')
        cases.check_pfft_vs_oracle(4, (32, 64, 48), 'D', grid=(-1,))
        cases.check_pfft_vs_oracle(4, (40, 64, 64), 'd', padding=[1.5, 1.5, 1.5])
    monkeypatch.setenv('GFFT_FUSE_PACK', '0')
    cases.check_pfft_vs_oracle(8, (64, 64, 64), 'D')


@pytest.mark.parametrize('P', [1, 2, 4])
def test_callers_output_array_is_written_directly(P):
    """forward(u, out) / backward(uh, out): the last kernel writes the caller's device array; the
    planned output array is not staged through (mpifft.py:75-77 copies instead)."""
    from mpi4py_fft_amd import PFFT, newDistArray
    shape = (48, 32, 40)
    ref = O.OPFFT(P, shape, dtype='d')
    G = O.rng_array(shape, 'd', 3)
    want = ref.forward(ref.scatter(G))

    def body(comm):
        fft = PFFT(comm, shape, dtype='d')
        u = newDistArray(fft, False)
        u[...] = G[fft.local_slice(False)]
        out = newDistArray(fft, True)
        planned = fft.forward.output_array
        planned.fill(7)
        assert fft.forward(u, out) is out
        untouched = bool(np.all(np.asarray(planned) == 7))
        back = newDistArray(fft, False)
        assert fft.backward(out, back) is back
        keep = np.asarray(out).copy()
        host = np.zeros(fft.shape(True), dtype='D')
        fft.forward(u, host)
        return keep, np.asarray(back).copy(), host, fft.local_slice(False), untouched
    for r, (uh, back, host, sl, untouched) in enumerate(cases.run_ranks(P, body)):
        assert untouched
        assert np.abs(uh - want[r]).max() < 1e-13 and np.abs(host - want[r]).max() < 1e-13
        assert np.abs(back - G[sl]).max() < 1e-13


@pytest.mark.parametrize('P', [1, 2])
def test_reference_docstring_example(P):
    """The usage example of the reference's PFFT docstring (mpifft.py:176-199), verbatim but for
    the import line: numpy shape array, np.zeros_like on a DistArray, numpy output arrays, and the
    DCT-III `transforms=` configuration."""
    import functools

    def body(comm):
        from mpi4py_fft_amd import PFFT, newDistArray
        N = np.array([12, 14, 15], dtype=int)
        fft = PFFT(comm, N, axes=(0, 1, 2))
        u = newDistArray(fft, False)
        u[:] = np.random.random(u.shape).astype(u.dtype)
        u_hat = fft.forward(u)
        uj = np.zeros_like(u)
        uj = fft.backward(u_hat, uj)
        assert np.allclose(uj, u)
        from mpi4py_fft_amd.fftw import rfftn, irfftn, dctn, idctn
        dct = functools.partial(dctn, type=3)
        idct = functools.partial(idctn, type=3)
        transforms = {(1, 2): (dct, idct)}
        r2c = PFFT(comm, N, axes=((0,), (1, 2)), transforms=transforms)
        u = newDistArray(r2c, False)
        u[:] = np.random.random(u.shape).astype(u.dtype)
        u_hat = r2c.forward(u)
        uj = np.zeros_like(u)
        uj = r2c.backward(u_hat, uj)
        assert np.allclose(uj, u)
        return True
    assert all(cases.run_ranks(P, body))


@pytest.mark.parametrize('P', [1, 2, 4])
def test_reference_docs_convolution_example(P):
    """docs/source/parallel.rst:329-354: alias-free convolution with the `padding` keyword, as
    written there (np.complex spelled complex), checked against the oracle's padded transforms."""
    N = (128, 128)
    ref = O.OPFFT(P, N, dtype='D', padding=[1.5, 1.5])
    rng = np.random.default_rng(0)
    A_hat = rng.random(ref.output_shape) + rng.random(ref.output_shape) * 1j
    B_hat = rng.random(ref.output_shape) + rng.random(ref.output_shape) * 1j
    ra = ref.backward(ref.scatter(A_hat, True))
    rb = ref.backward(ref.scatter(B_hat, True))
    want = ref.forward([x * y for x, y in zip(ra, rb)])

    def body(comm):
        from mpi4py_fft_amd import PFFT, newDistArray
        fft = PFFT(comm, N, padding=[1.5, 1.5], dtype=complex)
        a_hat = newDistArray(fft, True)
        b_hat = newDistArray(fft, True)
        a_hat[:] = A_hat[fft.local_slice(True)]
        b_hat[:] = B_hat[fft.local_slice(True)]
        a = newDistArray(fft, False)
        b = newDistArray(fft, False)
        assert a.shape == (192 // comm.Get_size(), 192)
        a = fft.backward(a_hat, a)
        b = fft.backward(b_hat, b)
        ab_hat = fft.forward(a * b)
        return np.asarray(ab_hat).copy()
    for r, got in enumerate(cases.run_ranks(P, body)):
        assert got.shape == want[r].shape
        assert np.abs(got - want[r]).max() <= 1e-12 * np.abs(want[r]).max()


@pytest.mark.gpu
@pytest.mark.parametrize('dt', 'dfDF')
@pytest.mark.parametrize('shape,padding', [((32, 32, 64), [1.5] * 3), ((64, 32, 128), [1.5] * 3),
                                           ((32, 64, 32), [1.5, 1, 2]), ((43, 32, 64), [1.5] * 3),
                                           ((128, 43, 64), [1.5, 1.5, 1.5]),      # (odd kept lengths: 43 -> 64)
                                           # round 5: 3/2-rule ONTO unequal-width stage lengths (160 -> 240, 224 -> 336, 320 -> 480)
                                           ((160, 224, 320), [1.5] * 3), ((80, 160, 160), [1.5, 1.5, 1.5])])
def test_padded_transform_as_one_plan(dt, shape, padding, monkeypatch):
    """One rank, padding: the three padded stages run as ONE plan through libgfft's pitched workspace
    (gfft_plan_create_padded).  Forward against the oracle (libfft.py:263-311 per stage), forward and
    backward against the staged chain of per-axis plans, and the reference's idempotence checks
    (tests/test_mpifft.py:229-251)."""
    from mpi4py_fft_amd import PFFT, comm, _lib
    from oracle import pfft_oracle as O
    _lib.set_option('fused3_min_mib', 0)
    monkeypatch.setenv('GFFT_PADDED_ONE_PLAN', 'fwd,bwd')       # (the default picks per direction by measurement)
    try:
        one = PFFT(comm.COMM_SELF, shape, dtype=dt, padding=list(padding))
        assert one._fused_plans is not None, 'the padded plan was not taken'
    finally:
        _lib.set_option('fused3_min_mib', 32)
    monkeypatch.delenv('GFFT_PADDED_ONE_PLAN')
    staged = PFFT(comm.COMM_SELF, shape, dtype=dt, padding=list(padding), fuse=False)
    assert staged._fused_plans is None
    ref = O.OPFFT(1, shape, dtype=dt, padding=list(padding))
    G = O.rng_array(ref.input_shape, dt, 11)
    want = ref.forward(ref.scatter(G))[0]
    from tests import cases
    tol = cases.tol_for(dt, G.size)                      # contract tolerance and rounding level, whichever is tighter
    u = np.asarray(G)
    uh = np.asarray(one.forward(u)).copy()
    assert uh.shape == want.shape
    assert np.abs(uh - want).max() <= tol * np.abs(want).max()
    uh_s = np.asarray(staged.forward(u)).copy()
    assert np.abs(uh - uh_s).max() <= tol * np.abs(uh_s).max()
    assert np.array_equal(np.asarray(one.forward.input_array), u), 'the input must be preserved'
    # backward of the same spectrum, both ways; then forward again (idempotence on the truncated space)
    b1 = np.asarray(one.backward(uh_s)).copy()
    b2 = np.asarray(staged.backward(uh_s)).copy()
    assert b1.shape == b2.shape == tuple(one.forward.input_array.shape)
    assert np.abs(b1 - b2).max() <= tol * max(np.abs(b2).max(), 1e-30)
    again = np.asarray(one.forward(b1))
    assert np.abs(again - uh_s).max() <= 10 * tol * np.abs(uh_s).max()
    # swapped normalisation (test_mpifft.py:240-248)
    one.backward.input_array[...] = uh_s
    one.backward(normalize=True)
    one.forward(normalize=False)
    assert np.abs(np.asarray(one.forward.output_array) - uh_s).max() <= 10 * tol * np.abs(uh_s).max()
    one.destroy()
    staged.destroy()


@pytest.mark.gpu
@pytest.mark.parametrize('dt', 'dD')
def test_padded_lengths_without_fused_adapters_take_the_staged_chain(dt):
    """27 -> 40 = 5 * 2^3: the shipped library has no fused truncation / zero-padding adapters for 5^c 2^k lengths
    (fused_pad_ok, kFusedPadMix5), so gfft_plan_create_padded answers UNSUPPORTED and PFFT keeps the staged chain
    with the separate gfft_truncate / gfft_pad kernels (libfft.py:263-311).  The path taken is asserted, and the
    result checked against the oracle."""
    from mpi4py_fft_amd import PFFT, comm, _lib
    from oracle import pfft_oracle as O
    shape, padding = (27, 32, 64), [1.5, 1.5, 1.5]
    _lib.set_option('fused3_min_mib', 0)
    try:
        fft = PFFT(comm.COMM_SELF, shape, dtype=dt, padding=list(padding))
    finally:
        _lib.set_option('fused3_min_mib', 32)
    assert fft._fused_plans is None, 'a 40-long padded axis cannot be part of the one-plan form in this build'
    by_axis = {x.axes[0]: x for x in fft.xfftn}
    assert by_axis[0]._padded and not by_axis[0]._fused_trunc           # 27 -> 40: stand-alone truncation / padding kernels
    assert by_axis[1]._fused_trunc and by_axis[2]._fused_trunc           # 32 -> 48, 64 -> 96: fused adapters
    ref = O.OPFFT(1, shape, dtype=dt, padding=list(padding))
    G = O.rng_array(ref.input_shape, dt, 12)
    want = ref.forward(ref.scatter(G))[0]
    uh = np.asarray(fft.forward(np.asarray(G))).copy()
    assert uh.shape == want.shape and np.abs(uh - want).max() <= cases_tol('D', G.size) * np.abs(want).max()
    back = np.asarray(fft.backward(uh))
    again = np.asarray(fft.forward(back))
    assert np.abs(again - uh).max() <= 2e-9 * np.abs(uh).max()
    fft.destroy()


@pytest.mark.gpu
def test_padded_distributed_transform_with_a_mix5_length_keeps_the_staged_wire():
    """The overlap pipeline needs every padded stage to carry fused adapters (pipeline.Pipeline.build); with a
    27 -> 40 axis it declines and the transform runs staged -- still against the oracle, on 4 thread-ranks."""
    from mpi4py_fft_amd import PFFT, newDistArray
    from oracle import pfft_oracle as O
    from tests import cases
    shape, padding, P = (27, 32, 64), [1.5, 1.5, 1.5], 4
    ref = O.OPFFT(P, shape, dtype='D', padding=list(padding))
    G = O.rng_array(ref.input_shape, 'D', 13)
    want = ref.forward(ref.scatter(G))

    def body(comm):
        fft = PFFT(comm, shape, dtype='D', padding=list(padding), wire='overlap', exchange='direct')
        piped = fft.pipeline is not None
        u = newDistArray(fft, False)
        u[...] = G[fft.local_slice(False)]
        uh = np.asarray(fft.forward(u)).copy()
        fft.destroy()
        return piped, uh
    for r, (piped, uh) in enumerate(cases.run_ranks(P, body)):
        assert not piped, 'a stage without fused adapters cannot be pipelined'
        assert uh.shape == want[r].shape and np.abs(uh - want[r]).max() <= cases_tol('D', G.size) * np.abs(want[r]).max()


@pytest.mark.parametrize('P,shape,dt,kw', [
    (4, (240, 112, 480), 'D', {}),                                  # 3 x 5 x 2^k / 7 x 2^k lengths on every axis, pencil grid
    (2, (240, 240, 224), 'd', {}),                                  # real: packed rows of 2 x 112, half spectrum 113 wide
    (8, (240, 240, 240), 'F', {}),
    (4, (160, 160, 160), 'D', dict(padding=[1.5, 1.5, 1.5])),       # 3/2-rule ONTO 240: truncation / padding on their own kernels
    (2, (240, 448), 'D', dict(collapse=True)),
    (4, (112, 240, 64), 'D', dict(grid=(-1,))),                     # slab
])
def test_unequal_width_stage_lengths_in_distributed_transforms(P, shape, dt, kw):
    """Round 5 lengths (csrc/fft_mixv_*.hip) inside multi-rank PFFTs: their plans refuse exchange-buffer layouts
    (gfft_plan_set_split / _set_tiles) and fused truncation, so these transforms take pack / unpack kernels, the staged
    wire and the stand-alone truncate / pad kernels -- same values as the oracle, same geometry as the reference."""
    cases.check_pfft_vs_oracle(P, shape, dt, **kw)
