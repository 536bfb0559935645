#!/usr/bin/env python3
"""Developer tool: pack/unpack kernel bandwidth on the BASELINE transfer shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mpi4py_fft_amd import _lib
from mpi4py_fft_amd.array import DeviceArray

eng = _lib.engine()


def timeit(fn, iters=5, warm=2):
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
for name, shape, axis, p, dt in (
        ('C4@8 T0 pack   (256,512,1024) axis2 /2 c128', (256, 512, 1024), 2, 2, 'D'),
        ('C4@8 T0 unpack (256,1024,512) axis1 /2 c128', (256, 1024, 512), 1, 2, 'D'),
        ('C4@8 T1 pack   (256,1024,512) axis1 /4 c128', (256, 1024, 512), 1, 4, 'D'),
        ('C4@2 T1 pack   (512,1024,1024) axis1 /2 c128', (512, 1024, 1024), 1, 2, 'D'),
        ('C5@8 T0 pack   (512,1024,1025) axis2 /2 c64', (512, 1024, 1025), 2, 2, 'F'),
        ('C5@8 T0 unpack (512,2048,513) axis1 /2 c64', (512, 2048, 513), 1, 2, 'F'),
        ('C5@8 T1 pack   (512,2048,513) axis1 /4 c64', (512, 2048, 513), 1, 4, 'F')):
    a = DeviceArray(shape, dt)
    b = DeviceArray(shape, dt)
    a.tensor.view(torch.float64 if dt == 'D' else torch.float32).normal_()
    isz = a.itemsize
    tp = timeit(lambda: eng.pack(a.tensor, b.tensor, shape, axis, p, isz))
    tu = timeit(lambda: eng.unpack(b.tensor, a.tensor, shape, axis, p, isz))
    tc = timeit(lambda: b.tensor.copy_(a.tensor))
    print('%-48s pack %7.3f ms %7.1f GB/s | unpack %7.3f ms %7.1f GB/s | copy %7.3f ms' % (
        name, tp, 2 * a.nbytes / tp / 1e6, tu, 2 * a.nbytes / tu / 1e6, tc), flush=True)
    del a, b
