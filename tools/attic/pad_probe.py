#!/usr/bin/env python3
"""Developer probe: one padded (3/2-rule) axis at a time, forward (truncating store) against
backward (zero-padding load), beside the unpadded pass of the same padded length."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from mpi4py_fft_amd import _lib
from mpi4py_fft_amd.libfft import FFT


def timeit(fn, iters=6, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(iters):
        s.record(); fn(); e.record(); e.synchronize()
        ts.append(s.elapsed_time(e))
    return min(ts)


def case(shape, axis, dt, pad):
    padding = [1.0] * len(shape)
    padding[axis] = pad
    f = FFT(shape, axes=(axis,), dtype=dt, padding=padding if pad != 1.0 else False)
    t = f.forward.input_array.tensor
    (torch.view_as_real(t) if t.is_complex() else t).normal_()
    tf = timeit(lambda: f.forward())
    tb = timeit(lambda: f.backward())
    fl, by, nl = f.fwd.cost()
    print('%-20s axis %d %s pad %.1f  fwd %7.3f ms %7.1f GB/s | bwd %7.3f ms %7.1f GB/s   out %s fused=%s' % (
        shape, axis, dt, pad, tf, by / tf / 1e6, tb, by / tb / 1e6, f.forward.output_array.shape, f._fused_trunc), flush=True)
    f.destroy()
    del f, t
    import gc
    gc.collect()
    torch.cuda.empty_cache()


print(torch.cuda.get_device_name(0))
if 'PAD_PROBE_VARIANT_ROWS' in os.environ:
    _lib.set_option('variant_rows', int(os.environ['PAD_PROBE_VARIANT_ROWS']))
for dt in 'DF':
    for axis in (2, 1, 0):
        shape = [1024, 1024, 1024] if dt == 'D' else [1024, 1024, 2048]
        case(tuple(shape), axis, dt, 1.0)
        case(tuple(shape), axis, dt, 1.5)
for dt in 'df':
    case((1024, 1024, 1024), 2, dt, 1.0)
    case((1024, 1024, 1024), 2, dt, 1.5)
    case((1024, 768, 1536), 2, dt, 1.0)
    case((1024, 768, 1536), 2, dt, 1.5)
for dt in 'DF':
    for axis in (2, 1, 0):
        case((768, 768, 768) if dt == 'D' else (768, 1536, 1536), axis, dt, 1.0)
        case((768, 768, 768) if dt == 'D' else (768, 1536, 1536), axis, dt, 1.5)
