#!/usr/bin/env python3
"""Developer probe: a strided pass over arrays whose rows are 512 / 513 / 520 / 528 wide (line-aligned
or not): how much of the half-spectrum penalty is misalignment, how much the ragged last tile."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import gc
import torch
from mpi4py_fft_amd import fftw, _lib
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


def case(shape, dt, axis, swz=None):
    a = DeviceArray(shape, dt)
    torch.view_as_real(a.tensor).normal_()
    if swz is not None:
        _lib.set_option('xcd_swizzle', swz)
    p = fftw.fftn(a, axes=(axis,))
    b = p.output_array
    t = timeit(lambda: p.execute_scaled(a, b, 1.0))
    fl, by, nl = p.cost()
    print('%-20s %s axis %d swizzle %-4s %8.3f ms %7.1f GB/s' % (shape, dt, axis, swz, t, by / t / 1e6), flush=True)
    p.destroy()
    _lib.set_option('xcd_swizzle', -1)
    del a, b, p
    gc.collect()
    torch.cuda.empty_cache()


print(torch.cuda.get_device_name(0))
if os.environ.get('WIDTH_PROBE_VARIANTS'):
    for v in [int(x) for x in os.environ['WIDTH_PROBE_VARIANTS'].split(',')]:
        _lib.set_option('variant_cols', v)
        print('variant_cols', v)
        for w in (512, 513):
            case((1024, 1024, w), 'D', 1, 1)
    sys.exit(0)
for w in (512, 513, 520, 528, 544):
    for swz in (0, 1):
        case((1024, 1024, w), 'D', 1, swz)
for w in (512, 513, 520, 528):
    for swz in (0, 1):
        case((1024, 1024, w), 'D', 0, swz)
for w in (512, 513, 520, 528):
    for swz in (0, 1):
        case((512, 2048, w), 'F', 1, swz)
for w in (512, 513, 520):
    for swz in (0, 1):
        case((2048, 512, w), 'F', 0, swz)
