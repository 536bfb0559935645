#!/usr/bin/env python3
"""Developer probe: one-rank padded PFFTs as one plan (gfft_plan_create_padded) against the staged
chain of per-axis plans with fused truncation."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mpi4py_fft_amd import PFFT, comm


def timeit(fn, iters=8, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(iters):
        s.record(); fn(); e.record(); e.synchronize()
        ts.append(s.elapsed_time(e))
    return min(ts)


print(torch.cuda.get_device_name(0))
for shape, dt in (((1024, 512, 512), 'f'), ((512, 512, 512), 'd'), ((683, 683, 683), 'D'), ((512, 512, 512), 'D'),
                  ((1024, 1024, 512), 'f')):
    r = []
    for which in ('fwd,bwd', 'none', None):
        os.environ.pop('GFFT_PADDED_ONE_PLAN', None)
        if which is not None:
            os.environ['GFFT_PADDED_ONE_PLAN'] = which
        f = PFFT(comm.COMM_SELF, shape, dtype=dt, padding=[1.5] * 3)
        (torch.view_as_real(f.forward.input_array.tensor) if dt in 'FD' else f.forward.input_array.tensor).normal_()
        tf, tb = timeit(f.forward), timeit(f.backward)
        _, nbytes = f.cost()
        r.append('%s fwd %.3f ms %5.0f GB/s  bwd %.3f ms %5.0f GB/s' % ({'fwd,bwd': 'one plan', 'none': 'staged', None: 'default'}[which], tf, nbytes / tf / 1e6, tb, nbytes / tb / 1e6))
        f.destroy()
    print(shape, dt, ' | '.join(r), flush=True)
