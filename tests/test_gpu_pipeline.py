"""The chunked, stream-overlapped redistribution (mpi4py-fft_amd/pipeline.py) on libgfft's own
communicators (C ABI gfft_comm_* / gfft_sendrecv), on ONE GPU: thread-ranks stand in for processes
and tests/fake_rccl stands in for RCCL, which refuses two ranks on a device.  Everything above the
dozen ncclXxx entry points is the product path: communicator split, message lists, streams and
events, guru plans addressing chunk-major exchange buffers."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from tests import cases, thread_comm
from oracle import pfft_oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope='module', autouse=True)
def fake_rccl():
    src = os.path.join(HERE, 'fake_rccl', 'fake_rccl.cpp')
    so = os.path.join(HERE, 'fake_rccl', 'libfake_rccl.so')
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(['/opt/rocm/bin/hipcc', '-O2', '-std=c++17', '-fPIC', '-shared', '-x', 'hip',
                               '--offload-arch=gfx950', src, '-o', so])
    from mpi4py_fft_amd import _lib
    _lib.check_wire(_lib.lib().gfft_rccl_load(so.encode()))
    yield
    _lib.lib().gfft_rccl_load(None)          # back to the real library for whoever comes next


def test_native_alltoallv_and_split_through_the_c_abi():
    """gfft_comm_create / gfft_comm_split / gfft_alltoallv with uneven counts on 4 thread-ranks."""
    import torch
    from mpi4py_fft_amd import _lib, comm

    def body(c):
        L = _lib.lib()
        w = comm.NativeWire.create(c)
        r, P = c.Get_rank(), c.Get_size()
        assert (w.rank, w.size) == (r, P)
        # rank r sends (r + j + 1) doubles to rank j, value 100 r + j
        scounts = [r + j + 1 for j in range(P)]
        rcounts = [j + r + 1 for j in range(P)]
        send = torch.cat([torch.full((n,), 100.0 * r + j, dtype=torch.float64) for j, n in enumerate(scounts)]).cuda()
        recv = torch.zeros(sum(rcounts), dtype=torch.float64, device='cuda')
        i64 = lambda v: (ctypes.c_int64 * len(v))(*v)
        sd = [sum(scounts[:j]) for j in range(P)]
        rd = [sum(rcounts[:j]) for j in range(P)]
        _lib.check_wire(L.gfft_alltoallv(w.handle, send.data_ptr(), i64(scounts), i64(sd), recv.data_ptr(),
                                         i64(rcounts), i64(rd), 8, _lib.current_stream()))
        torch.cuda.synchronize()
        want = torch.cat([torch.full((n,), 100.0 * j + r, dtype=torch.float64) for j, n in enumerate(rcounts)])
        assert torch.equal(recv.cpu(), want)
        # split into rows of a 2 x 2 grid: color = row, key = column
        sub = w.split(r // 2, r % 2, (2 * (r // 2), 2 * (r // 2) + 1))
        assert (sub.rank, sub.size) == (r % 2, 2)
        a = torch.full((6,), float(r), device='cuda')
        b = torch.empty(6, device='cuda')
        sub.alltoall_blocks(a.data_ptr(), b.data_ptr(), 12, _lib.current_stream().value or 0)
        torch.cuda.synchronize()
        row = [2 * (r // 2), 2 * (r // 2) + 1]
        assert b.cpu().tolist() == [float(row[0])] * 3 + [float(row[1])] * 3
        return True
    assert all(thread_comm.run(4, body))


@pytest.mark.parametrize('P,shape,kw', [
    (2, (64, 64, 64), {}),
    (4, (64, 64, 64), {}),
    (8, (64, 64, 64), {}),
    (8, (128, 64, 256), {}),
    (4, (64, 128, 64), dict(grid=(-1,))),          # slab: one redistribution over all ranks
    (8, (64, 64, 128), dict(grid=(-1,))),
    (4, (64, 64, 64), dict(axes=(1, 2, 0))),
    (8, (32, 32, 1024), {}),                       # long enough rows for tile-major exchange buffers in fp32 too
    (4, (32, 64, 512), {}),
    (2, (32, 32, 512), {}),
])
@pytest.mark.parametrize('dt', ['D', 'F'])
def test_pipelined_transform_is_bit_identical_to_the_staged_one(P, shape, kw, dt, monkeypatch):
    from mpi4py_fft_amd import PFFT, newDistArray, pipeline
    monkeypatch.setattr(pipeline.Pipeline, 'MIN_CHUNK_BYTES', 0)
    monkeypatch.setattr(pipeline.Pipeline, 'MIN_WIDTH', 4)
    G = O.rng_array(shape, dt, 11)

    def body(comm):
        k = {a: (list(b) if isinstance(b, list) else b) for a, b in kw.items()}
        staged = PFFT(comm, shape, dtype=dt, wire='torch', exchange='direct', **k)
        piped = PFFT(comm, shape, dtype=dt, wire='native', exchange='direct', **k)
        assert staged.pipeline is None and piped.pipeline is not None
        info = piped.pipeline.describe()
        u = newDistArray(staged, False)
        u[...] = G[staged.local_slice(False)]
        a = np.asarray(staged.forward(u)).copy()
        b = np.asarray(piped.forward(u)).copy()
        out = newDistArray(piped, True)
        piped.forward(u, out)                          # caller's arrays read / written directly
        c = np.asarray(out).copy()
        ab = np.asarray(staged.backward()).copy()
        bb = np.asarray(piped.backward()).copy()
        bn = np.asarray(piped.backward(out, normalize=True)).copy()
        keep = np.asarray(u).copy()
        staged.destroy()
        piped.destroy()
        return a, b, c, ab, bb, bn, keep, info
    res = cases.run_ranks(P, body)
    ref = O.OPFFT(P, shape, dtype=dt, **{k: (list(v) if isinstance(v, list) else v) for k, v in kw.items()})
    want = ref.forward(ref.scatter(G))
    for r, (a, b, c, ab, bb, bn, keep, info) in enumerate(res):
        assert any(e['chunks'] > 1 for e in info), info
        if shape[2] >= 512:
            assert all(e['layout'] == 'aligned' for e in info), info      # (pipeline._Aligned)
        assert np.array_equal(a, b) and np.array_equal(a, c), (P, shape, kw, r)
        assert np.array_equal(ab, bb)
        tol = cases.tol_for(dt, G.size)
        assert np.abs(a - want[r]).max() <= tol * np.abs(want[r]).max()
        assert np.allclose(bn * G.size, bb, rtol=1e-5 if dt == 'F' else 1e-12, atol=0)
    # the golden fixture of the reference itself through the pipelined path
    if P == 8 and shape == (64, 64, 64) and dt == 'D':
        monkeypatch.setenv('GFFT_WIRE', 'native')
        cases.check_pfft_golden('c2c_16x16x16_p8')


def test_transforms_that_do_not_qualify_keep_the_staged_path(monkeypatch):
    from mpi4py_fft_amd import PFFT

    def body(comm):
        out = []
        for shape, dt, kw in (((64, 64, 66), 'd', {}), ((48, 64, 60), 'D', {}),
                              ((60, 64, 64), 'D', dict(padding=[1.5, 1.5, 1.5])), ((32, 32), 'D', {})):
            f = PFFT(comm, shape, dtype=dt, wire='native', **kw)
            out.append(f.pipeline is None)
            f.destroy()
        return out
    monkeypatch.setenv('GFFT_WIRE', 'native')
    assert thread_comm.run(4, body)[0] == [True, True, True, True]
    cases.check_pfft_vs_oracle(4, (33, 20, 18), 'd')
    cases.check_pfft_vs_oracle(4, (48, 40, 64), 'D')


def test_one_rank_without_a_pipeline_keeps_every_rank_on_the_staged_path(monkeypatch):
    """Pipeline.build is local planning and may fail on a single rank (a local shape one of its
    stage plans cannot take); the grid then agrees on the staged path instead of leaving that
    rank's peers waiting in the pipelined exchange."""
    from mpi4py_fft_amd import PFFT, newDistArray, pipeline
    monkeypatch.setattr(pipeline.Pipeline, 'MIN_CHUNK_BYTES', 0)
    monkeypatch.setattr(pipeline.Pipeline, 'MIN_WIDTH', 4)
    build = pipeline.Pipeline.build.__func__

    def flaky(cls, pfft, wires, exchange=None, layout=None):
        pipe = build(cls, pfft, wires, exchange, layout)
        parent = next(c.relay_parent for c in pfft.subcomm if getattr(c, 'relay_parent', None) is not None)
        if pipe is not None and parent.Get_rank() == 1:
            pipe.destroy()
            return None
        return pipe
    monkeypatch.setattr(pipeline.Pipeline, 'build', classmethod(flaky))
    shape = (64, 64, 64)
    G = O.rng_array(shape, 'D', 3)

    def body(comm):
        f = PFFT(comm, shape, dtype='D', wire='native', exchange='relay')
        u = newDistArray(f, False)
        u[...] = G[f.local_slice(False)]
        a = np.asarray(f.forward(u)).copy()
        b = np.asarray(f.backward()).copy()
        piped = f.pipeline is not None
        sl = f.local_slice(False)
        f.destroy()
        return piped, a, b, sl
    res = cases.run_ranks(4, body)
    ref = O.OPFFT(4, shape, dtype='D')
    want = ref.forward(ref.scatter(G))
    for r, (piped, a, b, sl) in enumerate(res):
        assert not piped
        assert np.abs(a - want[r]).max() <= 1e-12 * np.abs(want[r]).max()
        assert np.allclose(b, G[sl], rtol=0, atol=1e-12)


def test_one_rank_without_the_aligned_layout_sends_everybody_back_to_c_order(monkeypatch):
    """Buffer layouts decide message sizes, so they are agreed on like the pipeline itself: a rank
    that cannot take the line-aligned exchange buffers reverts the whole grid to C-order ones."""
    from mpi4py_fft_amd import PFFT, newDistArray, pipeline
    monkeypatch.setattr(pipeline.Pipeline, 'MIN_CHUNK_BYTES', 0)
    monkeypatch.setattr(pipeline.Pipeline, 'MIN_WIDTH', 4)
    build = pipeline.Pipeline.build.__func__

    def picky(cls, pfft, wires, exchange=None, layout=None):
        parent = next(c.relay_parent for c in pfft.subcomm if getattr(c, 'relay_parent', None) is not None)
        return build(cls, pfft, wires, exchange, 'c-order' if parent.Get_rank() == 2 else layout)
    monkeypatch.setattr(pipeline.Pipeline, 'build', classmethod(picky))
    shape = (64, 64, 256)
    G = O.rng_array(shape, 'D', 4)

    def body(comm):
        f = PFFT(comm, shape, dtype='D', wire='native', exchange='direct')
        u = newDistArray(f, False)
        u[...] = G[f.local_slice(False)]
        a = np.asarray(f.forward(u)).copy()
        b = np.asarray(f.backward()).copy()
        lay = f.pipeline.layout
        sl = f.local_slice(False)
        f.destroy()
        return lay, a, b, sl
    res = cases.run_ranks(4, body)
    ref = O.OPFFT(4, shape, dtype='D')
    want = ref.forward(ref.scatter(G))
    for r, (lay, a, b, sl) in enumerate(res):
        assert lay == 'c-order'
        assert np.abs(a - want[r]).max() <= 1e-12 * np.abs(want[r]).max()
        assert np.allclose(b, G[sl], rtol=0, atol=1e-12)


@pytest.mark.parametrize('P,shape', [(4, (64, 64, 64)), (8, (64, 64, 64)), (8, (128, 64, 256))])
@pytest.mark.parametrize('chunks', [1, 4])
def test_pipelined_routed_exchange_is_bit_identical(P, shape, chunks, monkeypatch):
    """exchange='relay' on the native wire: every redistribution inside a small sub-communicator
    travels over all links of the grid in two rounds, round 2 of chunk k batched with round 1 of
    chunk k+1 (pipeline.Pipeline._plan_relays); same bits as the staged direct path."""
    from mpi4py_fft_amd import PFFT, newDistArray, pipeline
    monkeypatch.setattr(pipeline.Pipeline, 'MIN_CHUNK_BYTES', 0)
    monkeypatch.setattr(pipeline.Pipeline, 'MIN_WIDTH', 4)
    monkeypatch.setattr(pipeline.Pipeline, 'CHUNKS', chunks)
    G = O.rng_array(shape, 'D', 5)

    def body(comm):
        staged = PFFT(comm, shape, dtype='D', wire='torch', exchange='direct')
        piped = PFFT(comm, shape, dtype='D', wire='native', exchange='relay')
        info = piped.pipeline.describe()
        u = newDistArray(staged, False)
        u[...] = G[staged.local_slice(False)]
        a = np.asarray(staged.forward(u)).copy()
        b = np.asarray(piped.forward(u)).copy()
        ab = np.asarray(staged.backward()).copy()
        bb = np.asarray(piped.backward()).copy()
        b2 = np.asarray(piped.forward(u)).copy()          # buffers and events are reusable
        staged.destroy()
        piped.destroy()
        return a, b, ab, bb, b2, info
    for a, b, ab, bb, b2, info in cases.run_ranks(P, body):
        assert [e['route'] for e in info if e['ranks'] > 1] == ['relay'] * sum(1 for e in info if e['ranks'] > 1), info
        assert np.array_equal(a, b) and np.array_equal(ab, bb) and np.array_equal(a, b2)


@pytest.mark.parametrize('P,shape', [(2, (64, 64, 64)), (4, (64, 64, 128)), (8, (64, 64, 64)), (8, (128, 64, 256))])
@pytest.mark.parametrize('dt', ['d', 'f'])
@pytest.mark.parametrize('exchange', ['direct', 'relay'])
def test_pipelined_r2c_transform(P, shape, dt, exchange, monkeypatch):
    """r2c / c2r through the pipeline (config C5 in small): packed-real rows run slab by slab and
    write the chunk-major exchange buffer of UNEVEN blocks (n/2 + 1 entries over p ranks) themselves;
    the first exchange is an all-to-all(v).  Same bits as the staged path."""
    from mpi4py_fft_amd import PFFT, newDistArray, pipeline
    monkeypatch.setattr(pipeline.Pipeline, 'MIN_CHUNK_BYTES', 0)
    monkeypatch.setattr(pipeline.Pipeline, 'MIN_WIDTH', 4)
    G = O.rng_array(shape, dt, 23)

    def body(comm):
        staged = PFFT(comm, shape, dtype=dt, wire='torch', exchange='direct')
        piped = PFFT(comm, shape, dtype=dt, wire='native', exchange=exchange)
        assert piped.pipeline is not None
        info = piped.pipeline.describe()
        u = newDistArray(staged, False)
        u[...] = G[staged.local_slice(False)]
        a = np.asarray(staged.forward(u)).copy()
        b = np.asarray(piped.forward(u)).copy()
        out = newDistArray(piped, True)
        piped.forward(u, out)
        c = np.asarray(out).copy()
        ab = np.asarray(staged.backward()).copy()
        bb = np.asarray(piped.backward()).copy()
        back = newDistArray(piped, False)
        piped.backward(out, back)
        bc = np.asarray(back).copy()
        staged.destroy()
        piped.destroy()
        return a, b, c, ab, bb, bc, info
    res = cases.run_ranks(P, body)
    ref = O.OPFFT(P, shape, dtype=dt)
    want = ref.forward(ref.scatter(G))
    for r, (a, b, c, ab, bb, bc, info) in enumerate(res):
        assert any(e['chunks'] > 1 for e in info), info
        if exchange == 'relay' and P > 2:
            assert any(e['route'] == 'relay' for e in info), info
        assert np.array_equal(a, b) and np.array_equal(a, c), (P, shape, dt, r)
        assert np.array_equal(ab, bb) and np.array_equal(ab, bc)
        assert np.abs(a - want[r]).max() <= cases.tol_for(dt, G.size) * np.abs(want[r]).max()


@pytest.mark.parametrize('P,shape,pad', [
    (4, (64, 64, 64), [1.5, 1.5, 1.5]),            # padded lengths 96 = 3 * 32: the R = 12 / 24 kernels
    (8, (64, 64, 64), [1.5, 1.5, 1.5]),
    (8, (128, 64, 256), [1.5, 1.5, 1.5]),
    (4, (64, 128, 64), [1.5, 1.0, 1.5]),           # a plain stage between two truncating ones
    (8, (64, 64, 128), [2.0, 2.0, 2.0]),           # power-of-two padded lengths
    (2, (64, 64, 64), [1.5, 1.5, 1.5]),
])
@pytest.mark.parametrize('dt', ['D', 'F', 'd', 'f'])
def test_pipelined_padded_transform(P, shape, pad, dt, monkeypatch):
    """padding= (3/2-rule, mpifft.py:247-257, libfft.py:263-311) through the pipeline: every stage's
    truncating store (forward) / zero-padding load (backward) is fused into its guru plan
    (gfft_plan_create_guru_padded), the truncated side being the exchange buffer -- equal blocks of the
    kept entries, or, behind the real first stage, the block rule's uneven blocks of the kept half
    spectrum.  Same bits as the staged path, which now fuses its pack / unpack sides on padded axes too
    (PFFT._fuse_packs), and as the staged path with pack / unpack kernels; within tolerance of the oracle."""
    from mpi4py_fft_amd import PFFT, newDistArray, pipeline
    monkeypatch.setattr(pipeline.Pipeline, 'MIN_CHUNK_BYTES', 0)
    monkeypatch.setattr(pipeline.Pipeline, 'MIN_WIDTH', 4)
    pshape = tuple(int(np.floor(n * f)) for n, f in zip(shape, pad))
    G = O.rng_array(pshape, dt, 31)

    def body(comm):
        kernels = PFFT(comm, shape, dtype=dt, padding=list(pad), wire='torch', exchange='direct', fuse_pack=False)
        staged = PFFT(comm, shape, dtype=dt, padding=list(pad), wire='torch', exchange='direct')
        piped = PFFT(comm, shape, dtype=dt, padding=list(pad), wire='native', exchange='direct')
        assert kernels.pipeline is None and staged.pipeline is None and piped.pipeline is not None
        fused = [(t.packedA, t.packedB) for t in staged.transfer if t.comm.Get_size() > 1]
        info = piped.pipeline.describe()
        u = newDistArray(staged, False)
        assert tuple(u.global_shape) == pshape
        u[...] = G[staged.local_slice(False)]
        k = np.asarray(kernels.forward(u)).copy()
        a = np.asarray(staged.forward(u)).copy()
        b = np.asarray(piped.forward(u)).copy()
        out = newDistArray(piped, True)
        piped.forward(u, out)
        c = np.asarray(out).copy()
        kb = np.asarray(kernels.backward()).copy()
        ab = np.asarray(staged.backward()).copy()
        bb = np.asarray(piped.backward()).copy()
        back = newDistArray(piped, False)
        piped.backward(out, back)
        bc = np.asarray(back).copy()
        for f in (kernels, staged, piped):
            f.destroy()
        return k, a, b, c, kb, ab, bb, bc, info, fused
    res = cases.run_ranks(P, body)
    ref = O.OPFFT(P, shape, dtype=dt, padding=list(pad))
    want = ref.forward(ref.scatter(G))
    for r, (k, a, b, c, kb, ab, bb, bc, info, fused) in enumerate(res):
        assert any(e['chunks'] > 1 for e in info), info
        assert all(fa and fb for fa, fb in fused), fused          # no pack / unpack kernel left on padded axes
        assert np.array_equal(a, b) and np.array_equal(a, c), (P, shape, pad, dt, r)
        assert np.array_equal(ab, bb) and np.array_equal(ab, bc)
        # (pack / unpack kernels instead of fused sides: other instantiations of the transform kernels,
        # whose multiply-adds the compiler may contract differently -- last-bit agreement)
        eps = 1e-13 if dt in 'dD' else 1e-5
        assert np.abs(k - a).max() <= eps * np.abs(a).max() and np.abs(kb - ab).max() <= eps * np.abs(ab).max()
        assert np.abs(a - want[r]).max() <= cases.tol_for(dt, G.size) * np.abs(want[r]).max()


@pytest.mark.parametrize('P,shape,kw', [
    (4, (64, 64, 64), dict(grid=(-1,), collapse=True)),          # slab: stages [(1, 2) as one 2-D transform] -> [0]
    (8, (64, 48, 64), dict(grid=(-1,), collapse=True)),
    (8, (128, 64, 256), dict(grid=(-1,), collapse=True)),
    (4, (64, 36, 50), dict(grid=(-1,), collapse=True)),          # lengths without register kernels, uneven blocks
    (4, (64, 64, 64), dict(collapse=True)),                      # pencil grid: nothing collapses, the plain chain
    (8, (64, 64, 128), dict(collapse=True)),
])
@pytest.mark.parametrize('dt', ['D', 'd', 'f'])
def test_pipelined_collapsed_transform(P, shape, kw, dt, monkeypatch):
    """collapse=True (mpifft.py:299-306) through the pipeline.  On slab grids the leading serial transform
    covers both undistributed axes, the redistribution's free axis among them: the stage then runs slab by
    slab along array axis 0 and chunk c of the exchange leaves while slab c + 1 is transformed
    (pipeline._SlabStage).  Same bits as the staged path, within tolerance of the oracle."""
    from mpi4py_fft_amd import PFFT, newDistArray, pipeline
    monkeypatch.setattr(pipeline.Pipeline, 'MIN_CHUNK_BYTES', 0)
    monkeypatch.setattr(pipeline.Pipeline, 'MIN_WIDTH', 4)
    G = O.rng_array(shape, dt, 41)

    def body(comm):
        staged = PFFT(comm, shape, dtype=dt, wire='torch', exchange='direct', **kw)
        piped = PFFT(comm, shape, dtype=dt, wire='native', exchange='direct', **kw)
        assert staged.pipeline is None and piped.pipeline is not None
        info, axes = piped.pipeline.describe(), piped.axes
        u = newDistArray(staged, False)
        u[...] = G[staged.local_slice(False)]
        a = np.asarray(staged.forward(u)).copy()
        b = np.asarray(piped.forward(u)).copy()
        out = newDistArray(piped, True)
        piped.forward(u, out)
        c = np.asarray(out).copy()
        ab = np.asarray(staged.backward()).copy()
        bb = np.asarray(piped.backward()).copy()
        back = newDistArray(piped, False)
        piped.backward(out, back)
        bc = np.asarray(back).copy()
        b2 = np.asarray(piped.forward(u)).copy()
        staged.destroy()
        piped.destroy()
        return a, b, c, ab, bb, bc, b2, info, axes
    res = cases.run_ranks(P, body)
    ref = O.OPFFT(P, shape, dtype=dt, **kw)
    want = ref.forward(ref.scatter(G))
    for r, (a, b, c, ab, bb, bc, b2, info, axes) in enumerate(res):
        assert any(e['chunks'] > 1 for e in info), info
        if 'grid' in kw:
            assert len(axes) == 2 and tuple(axes[1]) == (1, 2), axes
        assert np.array_equal(a, b) and np.array_equal(a, c) and np.array_equal(a, b2), (P, shape, kw, dt, r)
        assert np.array_equal(ab, bb) and np.array_equal(ab, bc)
        assert np.abs(a - want[r]).max() <= cases.tol_for(dt, G.size) * np.abs(want[r]).max()


def test_late_messages_are_waited_for():
    """The pipelined path with every message of the wire landing 400 us late (FAKE_RCCL_DELAY_US,
    tests/fake_rccl): c2c and r2c, direct and routed, 2 and 4 chunks against the oracle -- and the
    negative control: with the arrival waits removed the same run must come out wrong (without the
    delay it does not: device copies of test-sized messages are over before the next kernel starts)."""
    import sys
    env = dict(os.environ, FAKE_RCCL_DELAY_US='400')
    worker = os.path.join(HERE, 'late_messages_worker.py')
    ok = subprocess.run([sys.executable, worker], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                        timeout=600, text=True)
    assert ok.returncode == 0 and 'results correct' in ok.stdout, ok.stdout[-2000:]
    neg = subprocess.run([sys.executable, worker, 'broken'], env=env, stdout=subprocess.PIPE,
                         stderr=subprocess.STDOUT, timeout=600, text=True)
    assert neg.returncode == 0 and 'results WRONG' in neg.stdout, neg.stdout[-2000:]


@pytest.mark.parametrize('P,shape,kw,dt', [
    (2, (64, 64, 64), {}, 'D'),
    (4, (64, 64, 64), {}, 'D'),
    (8, (64, 64, 128), {}, 'D'),
    (8, (32, 32, 1024), {}, 'D'),                  # line-aligned exchange buffers (pipeline._Aligned)
    (4, (64, 128, 64), dict(grid=(-1,)), 'D'),     # slab
    (4, (64, 64, 128), {}, 'd'),                   # the reference's default dtype: uneven 33 | 32 half-spectrum blocks
    (8, (32, 32, 1024), {}, 'd'),                  # ... on line-aligned buffers (513 = 512 + 1)
    (2, (64, 64, 64), {}, 'F'),
])
def test_admission_gates_on_every_wire_and_route(P, shape, kw, dt, monkeypatch):
    """bench.py's admission gates (mpi4py-fft_amd/selftest.py) on the HIP engine: the positional exchange check of
    every Transfer -- packed sides written / read by the neighbouring kernels included -- and of every chunk exchange
    of the pipeline, the forward against the DFT by definition, and word-for-word identity of the plans, on the staged
    and the pipelined wire, direct and relayed; then a relayed route with two pieces swapped (and swapped back on the
    way home, so that the round trip holds) must be named by the check."""
    import torch
    import bench
    from mpi4py_fft_amd import PFFT, newDistArray, pipeline, selftest
    monkeypatch.setattr(pipeline.Pipeline, 'MIN_CHUNK_BYTES', 0)
    monkeypatch.setattr(pipeline.Pipeline, 'MIN_WIDTH', 4)
    G = O.rng_array(shape, dt, 5)

    def body(comm):
        k = {a: (list(b) if isinstance(b, list) else b) for a, b in kw.items()}
        out = []
        prints = None
        for wire, exchange in (('torch', 'direct'), ('torch', 'relay'), ('native', 'direct'), ('native', 'relay')):
            f = PFFT(comm, shape, dtype=dt, wire=wire, exchange=exchange, **k)
            chk = selftest.exchange_check(f, comm)
            u = newDistArray(f, False)
            u[...] = G[f.local_slice(False)]
            u0 = u.tensor.clone()
            uh = f.forward(u).tensor
            torch.cuda.synchronize()
            err = selftest.forward_gate(f, comm, u0, uh)
            fp = comm.allgather_obj(selftest.fingerprint(uh))
            same = prints is None or fp == prints
            prints = fp
            packed = [(t.packedA, t.packedB) for t in f.transfer if t.comm.Get_size() > 1]
            routes = [t.exchange for t in f.transfer if t.comm.Get_size() > 1]
            bad = None
            if wire == 'torch' and exchange == 'relay' and any(comm.allgather_obj(bench.misroute(f))):
                back = np.asarray(f.backward(f.forward(u))).copy()
                rt = float(np.abs(back - G[f.local_slice(False)]).max())
                bad = (rt, selftest.exchange_check(f, comm), selftest.forward_gate(f, comm, u0, f.forward(u).tensor))
            out.append((wire, exchange, chk, err, same, packed, routes, f.pipeline is not None, bad))
            f.destroy()
        return out
    res = cases.run_ranks(P, body)
    for r, rows in enumerate(res):
        for wire, exchange, chk, err, same, packed, routes, piped, bad in rows:
            assert chk['result'] == 'bit-exact', (r, wire, exchange, chk)
            assert (chk.get('pipeline_chunk_exchanges', 0) > 0) == piped
            assert err <= (2e-10 if dt in 'dD' else 2e-4) and same, (r, wire, exchange, err, same)
            if wire == 'torch':
                assert any(a or b for a, b in packed), packed        # the kernels' own exchange-buffer layouts were checked
            if bad is not None:
                rt, chk2, err2 = bad
                assert rt < (1e-12 if dt in 'dD' else 1e-5)         # the round trip cannot see it
                assert chk2['result'] == 'FAILED' and 'misplaced' in chk2['failures'][0], chk2
                assert err2 > 1e-3                                   # ... the forward gate can
    if P > 2 and not kw and dt == 'D':
        assert any(row[8] is not None for row in res[0])
