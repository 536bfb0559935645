#!/usr/bin/env python3
"""Developer probe: strided single-axis c2c passes of 3^b 2^k lengths (what padded transforms run),
fp32 and fp64, aligned and 513-style row widths, per plan variant."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
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


VARIANTS = [int(v) for v in os.environ.get('MIX_PROBE_VARIANTS', '0').split(',')]


def case(shape, dt, axis):
    a = DeviceArray(shape, dt)
    torch.view_as_real(a.tensor).normal_()
    b = DeviceArray(shape, dt)
    for v in VARIANTS:
        _lib.set_option('variant_cols', v)
        p = fftw.fftn(a, axes=(axis,), output_array=b)
        t = timeit(lambda: p.execute_scaled(a, b, 1.0))
        want = torch.fft.fft(a.tensor[:2, :2].to(torch.complex128), dim=axis) if axis == 2 else None
        sl = (slice(None), 0, 0) if axis == 0 else (0, slice(None), 0)
        want = torch.fft.fft(a.tensor[sl].to(torch.complex128))
        err = float((b.tensor[sl].to(torch.complex128) - want).abs().max() / want.abs().max())
        fl, by, nl = p.cost()
        print('%-18s %s axis %d variant %d  %7.3f ms %7.1f GB/s err %.1e  %s' % (
            shape, dt, axis, v, t, by / t / 1e6, err, p._eng.plan_describe(p._plan).splitlines()[1].strip()[:50]), flush=True)
        p.destroy()
    _lib.set_option('variant_cols', 0)


print(torch.cuda.get_device_name(0))
if os.environ.get('MIX_PROBE_SET') == 'pow2f':
    for shape, axis in [((1024, 1024, 1024), 1), ((1024, 1024, 1024), 0), ((1024, 512, 512), 0), ((512, 1024, 512), 1),
                        ((512, 512, 1024), 1), ((512, 512, 1024), 0), ((2048, 512, 513), 0), ((512, 1024, 513), 1)]:
        case(shape, 'F', axis)
        torch.cuda.empty_cache()
    sys.exit(0)
for dt in 'FD':
    for shape, axis in [((1536, 512, 512), 0), ((1536, 512, 513), 0), ((512, 1536, 512), 1), ((512, 1536, 513), 1),
                        ((768, 1024, 512), 0), ((1024, 768, 512), 1), ((1024, 768, 257), 1),
                        ((1024, 512, 512), 0), ((512, 1024, 512), 1), ((2048, 512, 512), 0), ((512, 2048, 512), 1)]:
        case(shape, dt, axis)
        torch.cuda.empty_cache()
