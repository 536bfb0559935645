#!/usr/bin/env python3
"""Developer probe: r2c / c2r along the contiguous axis -- packed-real half-length kernels
(fft_real_*.hip) against the full-length form, per variant; checks values against torch.fft."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from mpi4py_fft_amd import fftw, _lib
from mpi4py_fft_amd.array import DeviceArray


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


VARIANTS = [int(v) for v in os.environ.get('REAL_PROBE_VARIANTS', '0,1').split(',')]


def case(shape, dt):
    a = DeviceArray(shape, dt)
    a.tensor.normal_()
    ref = None
    for half, variant in [(0, 0)] + [(1, v) for v in VARIANTS]:
        _lib.set_option('real_half', half)
        _lib.set_option('variant_rows', variant)
        p = fftw.rfftn(a, axes=(len(shape) - 1,))
        b = p.output_array
        q = fftw.irfftn(b, s=(shape[-1],), axes=(len(shape) - 1,), output_array=DeviceArray(shape, dt))
        c = q.output_array
        tf = timeit(lambda: p.execute_scaled(a, b, 1.0))
        want = torch.fft.rfft(a.tensor[:2].to(torch.float64), dim=-1)
        err = float((b.tensor[:2].to(torch.complex128) - want).abs().max() / want.abs().max())
        tb = timeit(lambda: q.execute_scaled(b, c, 1.0 / shape[-1]))
        rt = float((c.tensor[:4] - a.tensor[:4]).abs().max())
        fl, by, nl = p.cost()
        print('%-22s %s half=%d variant=%d  r2c %7.3f ms %7.1f GB/s err %.1e | c2r %7.3f ms %7.1f GB/s rt %.1e  %s' % (
            shape, dt, half, variant, tf, by / tf / 1e6, err, tb, by / tb / 1e6, rt,
            p._eng.plan_describe(p._plan).splitlines()[1].strip()[:60]), flush=True)
        p.destroy(); q.destroy()
        del b, c, p, q
        torch.cuda.empty_cache()
    _lib.set_option('real_half', 1)
    _lib.set_option('variant_rows', 0)


print(torch.cuda.get_device_name(0))
if os.environ.get('REAL_PROBE_SET') == 'tune':
    case((1024, 1024, 1024), 'f')
    case((512, 1024, 2048), 'f')
    case((256, 1024, 4096), 'f')
    case((1024, 1024, 1024), 'd')
    case((512, 1024, 2048), 'd')
    case((256, 1024, 4096), 'd')
else:
    case((64, 64, 64), 'd')
    case((1024, 1024, 1024), 'd')
    case((1024, 1024, 1024), 'f')
    case((512, 1024, 2048), 'f')
    case((512, 1024, 2048), 'd')
    case((1024, 1024, 512), 'd')
    case((256, 1024, 4096), 'f')
    case((128, 1024, 8192), 'f')
    case((4096, 256), 'd')
    case((768, 768, 768), 'd')
    case((512, 768, 1536), 'f')
    case((1000, 1000, 1000), 'd')
    case((512, 1024, 3072), 'f')
