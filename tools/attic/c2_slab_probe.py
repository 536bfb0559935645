#!/usr/bin/env python3
"""Developer probe: batched 2^20-point transforms run group by group (G signals per four-step
execution), so that the intermediate of one group can stay in the 256 MiB Infinity Cache."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mpi4py_fft_amd import fftw
from mpi4py_fft_amd.array import DeviceArray


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
for dt, B in (('D', 64), ('F', 128)):
    N = 1 << 20
    a = DeviceArray((B, N), dt)
    torch.view_as_real(a.tensor).normal_()
    b = DeviceArray((B, N), dt)
    for G in (B, 32, 16, 8, 4, 2, 1):
        if G > B:
            continue
        sub = DeviceArray((G, N), dt, tensor=a.tensor[:G])
        osub = DeviceArray((G, N), dt, tensor=b.tensor[:G])
        p = fftw.fftn(sub, axes=(1,), output_array=osub)
        views = [(DeviceArray((G, N), dt, tensor=a.tensor[g:g + G]), DeviceArray((G, N), dt, tensor=b.tensor[g:g + G]))
                 for g in range(0, B, G)]

        def run():
            for x, y in views:
                p.execute_scaled(x, y, 1.0)
        t = timeit(run)
        esz = 16 if dt == 'D' else 8
        print('%s B=%d groups of %3d: %7.3f ms  %7.1f GB/s algorithmic (2 S)' % (dt, B, G, t, 2 * B * N * esz / t / 1e6), flush=True)
        p.destroy()
