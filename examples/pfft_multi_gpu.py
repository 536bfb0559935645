#!/usr/bin/env python3
"""PFFT on several MI355X GPUs of one node, one process per GPU.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \\
        examples/pfft_multi_gpu.py --size 1024 --dtype D [--grid 8] [--wire native|overlap|torch] [--exchange direct|relay|auto]

The script is the reference's usage pattern (mpi4py_fft: `fft = PFFT(MPI.COMM_WORLD, N, dtype=...)`,
`u = newDistArray(fft, False)`, `u_hat = fft.forward(u)`, `fft.backward(u_hat)`) with the world
communicator taken from `comm.init_distributed()` instead of mpi4py.  It checks the round trip and
one analytic mode, then times forward + backward.

  --wire      how a redistribution travels: `native` = chunks on libgfft's own RCCL communicators,
              overlapped with the serial transforms (pipeline.py); `overlap` = the same pipeline on
              asynchronous torch.distributed all-to-alls; `torch` = one all-to-all per redistribution
  --exchange  `relay` routes exchanges inside small sub-communicators over all xGMI links of the grid
  --grid      e.g. `8` for a slab decomposition on 8 ranks (one exchange over all links); default: the
              reference's pencil grid (Compute_dims)
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np
import torch

from mpi4py_fft_amd import PFFT, newDistArray, comm


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--size', type=int, default=512)
    ap.add_argument('--dtype', default='D', choices=list('dDfF'))
    ap.add_argument('--grid', type=int, nargs='*', default=None)
    ap.add_argument('--wire', default=None)
    ap.add_argument('--exchange', default=None)
    ap.add_argument('--steps', type=int, default=10)
    args = ap.parse_args()

    world = comm.init_distributed()
    rank, size = world.Get_rank(), world.Get_size()
    n = args.size
    kw = {}
    if args.grid:
        kw['grid'] = tuple(args.grid)
    fft = PFFT(world, (n, n, n), dtype=args.dtype, wire=args.wire, exchange=args.exchange, **kw)
    u = newDistArray(fft, False)
    # a plane wave: exactly one spectral coefficient (3, 5, 7) of amplitude 1 (1/2 for real data)
    sl = fft.local_slice(False)
    x = [torch.arange(s.start, s.stop, device='cuda', dtype=torch.float64) for s in sl]
    phase = 2 * np.pi * (3 * x[0][:, None, None] + 5 * x[1][None, :, None] + 7 * x[2][None, None, :]) / n
    if args.dtype in 'DF':
        u.tensor.copy_(torch.polar(torch.ones_like(phase), phase).to(u.tensor.dtype))
    else:
        u.tensor.copy_(torch.cos(phase).to(u.tensor.dtype))
    u0 = u.tensor.clone()
    uh = fft.forward(u)
    so = fft.local_slice(True)
    peak = 0.0
    if all(s.start <= k < s.stop for s, k in zip(so, (3, 5, 7))):
        peak = abs(complex(uh.tensor[3 - so[0].start, 5 - so[1].start, 7 - so[2].start].item()))
    peak = max(world.allgather_obj(peak))
    back = fft.backward(uh)
    err = float((back.tensor - u0).abs().max().item())
    err = max(world.allgather_obj(err))
    for _ in range(2):
        fft.backward(fft.forward(u))
    world.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        fft.backward(fft.forward(u))
    torch.cuda.synchronize()
    world.barrier()
    dt = world.allreduce_max(time.perf_counter() - t0) / args.steps
    if rank == 0:
        grid = [c.Get_size() for c in fft.subcomm]
        plan = fft.pipeline.describe() if fft.pipeline is not None else [dict(ranks=t.comm.Get_size(), route=t.exchange) for t in fft.transfer]
        flops = 2 * 5 * n ** 3 * np.log2(float(n) ** 3) * (1.0 if args.dtype in 'DF' else 0.5)
        print('%d^3 %s on %d GPU(s), grid %s: peak |u_hat[3,5,7]| = %.6f, round trip max err %.2e, '
              '%.2f ms per forward+backward = %.1f GFLOP/s; redistribution plan %s'
              % (n, args.dtype, size, grid, peak, err, dt * 1e3, flops / dt / 1e9, plan))
    fft.destroy()
    world.barrier()
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
