"""Worker for tests/test_gloo_distributed.py: one process per rank over torch.distributed (gloo).
Exercises the REAL communicator path (comm.TorchComm: Cartesian sub-groups via new_group,
all_to_all_single with uneven splits) under the product PFFT / Transfer / DistArray classes.
The arithmetic comes from the CPU checker engine (tests/host_engine.py) because this container has
no GPU; on GPUs the same code runs with the HIP engine over the nccl (RCCL) backend."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np


def main():
    from mpi4py_fft_amd import comm, _lib, PFFT, newDistArray, DistArray, Subcomm, selftest
    import bench
    from tests.host_engine import HostEngine
    from oracle import pfft_oracle as O
    _lib.set_engine(HostEngine())
    world = comm.init_distributed('gloo')
    P, r = world.Get_size(), world.Get_rank()
    from mpi4py_fft_amd import pencil
    pencil.Transfer.CHUNK_MIN_BYTES = 0     # exercise the chunked asynchronous exchange over gloo
    pencil.Transfer.CHUNKS = 3
    pencil.Transfer.RELAY_MIN_BYTES = 0     # measure the routes even on these tiny arrays
    cases = [((16, 12, 10), "D", {}), ((13, 12, 10), "d", {}), ((7, 8, 9), "D", {}),
             ((12, 13), 'D', {}), ((16, 12, 10), 'd', dict(padding=[1.5, 1.5, 1.5])),
             ((12, 10, 8), 'D', dict(grid=(-1,))), ((12, 9, 8, 6), 'd', dict(axes=((0,), (1,), (2, 3))))]
    from mpi4py_fft_amd import relay
    relayed, run = [], relay.Schedule.run
    misrouted = []
    relay.Schedule.run = lambda self, *a: (relayed.append(1), run(self, *a))[1]
    # second sweep: the two-round multi-path exchange (relay.py) over point-to-point messages
    sweeps = [(c, '0') for c in cases] + ([(c, '1') for c in cases] if P > 2 else [])
    for (shape, dt, kw), mode in sweeps:
        os.environ['GFFT_RELAY'] = mode if (mode == '0' or shape[0] != 7) else 'measure'
        ref = O.OPFFT(P, shape, dtype=dt, **{k: (list(v) if isinstance(v, list) else v) for k, v in kw.items()})
        G = O.rng_array(ref.input_shape, dt, 42)
        want = ref.forward(ref.scatter(G))[r]
        fft = PFFT(world, shape, dtype=dt, **{k: (list(v) if isinstance(v, list) else v) for k, v in kw.items()})
        assert [c.Get_size() for c in fft.subcomm] == list(ref.dims), (shape, kw)
        # what bench.py admits a plan on: every hop of every Transfer positional and bit-exact on this route
        chk = selftest.exchange_check(fft, world)
        assert chk['result'] == 'bit-exact' and chk['hops'] == 2 * len(fft.transfer), (shape, dt, kw, mode, chk)
        u = newDistArray(fft, False)
        assert tuple(u.shape) == ref.pencil_in[r].subshape
        u[...] = G[fft.local_slice(False)]
        uh = np.asarray(fft.forward(u))
        assert uh.shape == want.shape, (uh.shape, want.shape)
        assert np.abs(uh - want).max() <= 1e-12 * max(1.0, np.abs(want).max()), (shape, dt, kw)
        if dt in 'Dd' and len(shape) == 3 and not kw.get('padding'):      # (complex, and real with the halved axis last)
            import torch
            u0 = torch.from_numpy(np.ascontiguousarray(G[fft.local_slice(False)]))
            assert selftest.forward_gate(fft, world, u0) <= 1e-13, (shape, kw, mode)
            if P > 1:
                # ... and the gate notices a block in the wrong place where a round trip does not
                bad = fft.forward.output_array.tensor.clone()
                bad.copy_(torch.roll(bad, 1, 0))
                assert selftest.forward_gate(fft, world, u0, uh=bad) > 1e-3
        if not kw.get('padding'):
            back = np.asarray(fft.backward())
            assert np.abs(back - G[fft.local_slice(False)]).max() < 1e-12
        if mode == '1' and any(t._relay for t in fft.transfer) and not kw.get('padding'):
            # a relay piece delivered to the wrong offset, mirrored in the backward schedule: the round trip
            # still passes, the positional check names it
            if any(world.allgather_obj(bench.misroute(fft))):
                fft.forward(u)
                back = np.asarray(fft.backward())
                assert np.abs(back - G[fft.local_slice(False)]).max() < 1e-12
                chk = selftest.exchange_check(fft, world)
                assert chk['result'] == 'FAILED' and 'misplaced' in chk['failures'][0], chk
                misrouted.append(1)
        fft.destroy()
    assert bool(relayed) == (P > 2) and bool(misrouted) == (P > 2)
    # the chunked pipeline (pipeline.py) on torch.distributed's asynchronous all-to-all: layouts,
    # chunk offsets, guru-plan geometry and exchange order, against the staged path and the oracle
    from mpi4py_fft_amd import pipeline
    pipeline.Pipeline.MIN_CHUNK_BYTES = 0
    pipeline.Pipeline.MIN_WIDTH = 2
    misplaced_chunks = []
    os.environ['GFFT_RELAY'] = '0'
    # (the 256-wide ones take the line-aligned exchange buffers of pipeline._Aligned: tile-major T0, pitched T1; complex
    # transforms on slab grids run their two local stages as one launch per chunk of planes, pipeline._PairStage --
    # and stage by stage with GFFT_FUSE_PAIRS=0)
    for shape, dt, kw, layout in (((32, 16, 64), 'D', {}, None), ((16, 32, 32), 'F', {}, None),
                                  ((32, 32, 16), 'D', dict(grid=(-1,)), 'slab-pair'),
                                  ((16, 32, 256), 'D', {}, 'aligned'), ((32, 16, 256), 'F', {}, 'aligned'),
                                  ((64, 32, 64), 'F', dict(grid=(-1,)), 'slab-pair'),
                                  ((32, 32, 64), 'D', dict(grid=(-1,), fuse_pairs=None), 'aligned')):
        kw = dict(kw)
        os.environ['GFFT_FUSE_PAIRS'] = '0' if 'fuse_pairs' in kw else '1'
        if P == 2 and layout == 'aligned' and 'fuse_pairs' not in kw:
            layout = 'slab-pair'                   # (two ranks: the default grid IS a slab)
        kw.pop('fuse_pairs', None)
        if any(n % 8 for n in shape) and P == 8:
            continue
        ref = O.OPFFT(P, shape, dtype=dt, **{k: (list(v) if isinstance(v, list) else v) for k, v in kw.items()})
        G = O.rng_array(shape, dt, 17)
        want = ref.forward(ref.scatter(G))[r]
        piped = PFFT(world, shape, dtype=dt, wire='overlap', **kw)
        assert piped.pipeline is not None, (shape, kw)
        assert layout is None or piped.pipeline.layout == layout, (shape, dt, kw, piped.pipeline.describe())
        chk = selftest.exchange_check(piped, world)
        assert chk['result'] == 'bit-exact' and chk['pipeline_chunk_exchanges'] > 0, (shape, dt, kw, chk)
        u = newDistArray(piped, False)
        u[...] = G[piped.local_slice(False)]
        uh = np.asarray(piped.forward(u))
        tol = 1e-12 if dt == 'D' else 1e-5
        assert np.abs(uh - want).max() <= tol * max(1.0, np.abs(want).max()), (shape, dt, kw, np.abs(uh - want).max())
        back = np.asarray(piped.backward())
        assert np.abs(back - G[piped.local_slice(False)]).max() <= 10 * tol
        # a wire that delivers the messages of chunk c to the places of chunk K-1-c: the pipeline's positional
        # check names the chunk and the peer
        e = next(t for t in piped.pipeline.tplan if t['p'] > 1)
        if e['K'] > 1 and 'peer_stride' not in e['B'] and 'peer_stride' not in e['A']:
            w, K = e['wire'], e['K']
            good = w.exchange_chunk
            w.exchange_chunk = lambda send, so, ss, recv, ro, rs: good(send, so, ss, recv, (K - 1) * sum(rs) - ro, rs)
            try:
                chk = selftest.exchange_check(piped, world)
                assert chk['result'] == 'FAILED' and 'words wrong' in chk['failures'][0], chk
                misplaced_chunks.append(1)
            finally:
                w.exchange_chunk = good
            assert selftest.exchange_check(piped, world)['result'] == 'bit-exact'
        piped.destroy()
    os.environ['GFFT_FUSE_PAIRS'] = '1'
    assert misplaced_chunks
    # DistArray.redistribute over the real communicator (tests/test_darray.py:50-57)
    N = (8, 10, 12)
    sub = Subcomm(world, [0, 0, 1])
    z = DistArray(N, subcomm=sub, dtype=float, alignment=2)
    z[...] = np.random.default_rng(r).random(z.shape)
    n0 = sum(world.allgather_obj(float(np.sum(np.asarray(z) ** 2))))
    z1 = z.redistribute(1)
    n1 = sum(world.allgather_obj(float(np.sum(np.asarray(z1) ** 2))))
    assert np.isclose(n0, n1)
    # storage boundary (io.py): the ranks take turns at one NetCDF file, then everybody reads it back
    import tempfile
    from mpi4py_fft_amd import NCFile
    from scipy.io import netcdf_file
    path = world.bcast(os.path.join(tempfile.mkdtemp(), 'snap.nc') if r == 0 else None)
    G = np.random.default_rng(5).standard_normal(N)
    z[...] = G[z.local_slice()]
    if r == 0:
        NCFile(path, mode='w')
    world.barrier()
    f = NCFile(path, mode='a')
    for step in (0, 1):
        f.write(step, {'u': [z, (z, [slice(None), 3, slice(None)])]})
    z1.write(path, 'u1', 0)                     # a differently aligned array of the same grid
    back = DistArray(N, subcomm=sub, dtype=float, alignment=2)
    back.read(path, 'u', 1)
    assert np.array_equal(np.asarray(back), G[z.local_slice()])
    world.barrier()
    if r == 0:
        nc = netcdf_file(path, 'r', mmap=False)
        assert np.array_equal(nc.variables['u'].data[1], G)
        assert np.array_equal(nc.variables['u_slice_3_slice'].data[0], G[:, 3, :])
        assert nc.variables['u1'].data.shape[1:] == N           # (records are shared: two of them)
        nc.close()
    world.barrier()
    if r == 0:
        print('GLOO_WORKER_OK ranks=%d' % P)
    import torch.distributed as dist
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
