"""Worker for tests/test_gpu_multiproc.py: several processes share ONE GPU and talk over
torch.distributed "gloo" with device tensors, so the product path (HIP kernels, comm.TorchComm
process groups, Transfer with device staging buffers, chunked asynchronous exchange) runs end to
end across real processes on a 1-GPU box.  (RCCL itself refuses two ranks on one device.)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch


def main():
    from mpi4py_fft_amd import comm, PFFT, newDistArray, pencil
    from oracle import pfft_oracle as O
    torch.cuda.set_device(0)
    world = comm.init_distributed('gloo')
    P, r = world.Get_size(), world.Get_rank()
    pencil.Transfer.CHUNK_MIN_BYTES = 1 << 16
    cases = [((64, 48, 40), 'D', {}), ((48, 64, 66), 'd', {}), ((96, 64, 32), 'F', dict(grid=(-1,))),
             ((32, 48, 64), 'd', dict(padding=[1.5, 1.5, 1.5])), ((128, 128, 128), 'D', {})]
    # second sweep: the two-round multi-path exchange (relay.py) over point-to-point messages
    sweeps = [(c, '0') for c in cases] + ([(c, '1') for c in cases] if P > 2 else [])
    for (shape, dt, kw), mode in sweeps:
        os.environ['GFFT_RELAY'] = mode
        ref = O.OPFFT(P, shape, dtype=dt, **{k: (list(v) if isinstance(v, list) else v) for k, v in kw.items()})
        G = O.rng_array(ref.input_shape, dt, 42)
        want = ref.forward(ref.scatter(G))[r]
        fft = PFFT(world, shape, dtype=dt, **{k: (list(v) if isinstance(v, list) else v) for k, v in kw.items()})
        u = newDistArray(fft, False)
        assert u.tensor.is_cuda
        u[...] = G[fft.local_slice(False)]
        uh = np.asarray(fft.forward(u))
        tol = 2e-10 if dt in 'dD' else 2e-4
        assert uh.shape == want.shape
        assert np.abs(uh - want).max() <= tol * max(1e-30, np.abs(want).max()), (shape, dt, kw)
        if not kw.get('padding'):
            back = np.asarray(fft.backward())
            assert np.abs(back - G[fft.local_slice(False)]).max() <= 100 * tol
        fft.destroy()
    # the chunked, stream-overlapped pipeline (pipeline.py) on torch.distributed's asynchronous
    # all-to-all: same bits as the staged path, across real processes
    from mpi4py_fft_amd import pipeline
    pipeline.Pipeline.MIN_CHUNK_BYTES = 0
    pipeline.Pipeline.MIN_WIDTH = 4
    os.environ['GFFT_RELAY'] = '0'
    for shape, dt, kw in (((64, 64, 64), 'D', {}), ((128, 64, 32), 'F', {}), ((64, 64, 128), 'D', dict(grid=(-1,))),
                          ((64, 64, 64), 'd', {}), ((64, 64, 128), 'f', {})):
        staged = PFFT(world, shape, dtype=dt, wire='torch', **kw)
        piped = PFFT(world, shape, dtype=dt, wire='overlap', **kw)
        assert piped.pipeline is not None, (shape, dt, kw)
        # (on two ranks a real transform's only redistribution runs along the ragged half-spectrum
        # axis -- 33 | 32 columns -- which no common chunk count divides: one chunk there)
        assert any(e['chunks'] > 1 for e in piped.pipeline.describe()) or (dt in 'df' and P == 2), piped.pipeline.describe()
        G = O.rng_array(shape, dt, 9)
        u = newDistArray(staged, False)
        u[...] = G[staged.local_slice(False)]
        a = np.asarray(staged.forward(u)).copy()
        b = np.asarray(piped.forward(u)).copy()
        assert np.array_equal(a, b), (shape, dt, kw, 'forward')
        assert np.array_equal(np.asarray(staged.backward()), np.asarray(piped.backward())), (shape, dt, kw, 'backward')
        b2 = np.asarray(piped.forward(u)).copy()
        assert np.array_equal(a, b2)
        staged.destroy()
        piped.destroy()
    world.barrier()
    if r == 0:
        print('GPU_MULTIPROC_OK ranks=%d' % P)
    import torch.distributed as dist
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
