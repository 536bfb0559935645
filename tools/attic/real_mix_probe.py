#!/usr/bin/env python3
"""Developer probe: packed-real rows of mixed (2^a 3^b) lengths, r2c / c2r, with and without the
half-spectrum truncation of padded transforms; values checked against torch.fft."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mpi4py_fft_amd import fftw
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


def case(shape, dt):
    a = DeviceArray(shape, dt)
    a.tensor.normal_()
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
    print('%-20s %s r2c %7.3f ms %7.1f GB/s err %.1e | c2r %7.3f ms %7.1f GB/s rt %.1e  %s' % (
        shape, dt, tf, by / tf / 1e6, err, tb, by / tb / 1e6, rt,
        p._eng.plan_describe(p._plan).splitlines()[1].strip()[:70]), flush=True)


print(torch.cuda.get_device_name(0))
for dt in 'fd':
    for shape in [(512, 1024, 3072), (512, 768, 1536), (1024, 1024, 768), (2048, 1024, 384),
                  (512, 1024, 1152), (256, 1024, 6144)]:
        case(shape, dt)
        torch.cuda.empty_cache()
