#!/usr/bin/env python3
"""Developer probe (round 6): what bounds the complex64 strided passes of config C5's stages -- the default kernels against the
ACCESS PATTERN ALONE (variant_cols 10 of a library whose fft_pow2_f32 was built with -DGFFT_VARIANTS: the same tiles loaded and
stored, no butterflies, no LDS exchange) and the streaming copy of the same bytes.
usage: GFFT_AB_LIB=libgfft_var.so python tools/strided_bound_probe_f32.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpi4py_fft_amd import _lib
if os.environ.get('GFFT_AB_LIB'):
    _lib.LIBPATH = os.path.join(os.path.dirname(_lib.LIBPATH), os.environ['GFFT_AB_LIB'])
import torch


def timeit(fn, iters=12, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(iters):
        s.record(); fn(); e.record(); e.synchronize()
        ts.append(s.elapsed_time(e))
    return min(ts)


eng = _lib.engine()
L, st = _lib.lib(), _lib.current_stream()
print(torch.cuda.get_device_name(0))
for name, shape, axis in (('C5 stage 1 (512,2048,512) axis 1', (512, 2048, 512), 1), ('C5 stage 2 (2048,512,512) axis 0', (2048, 512, 512), 0),
                          ('(512,1024,1024) axis 1', (512, 1024, 1024), 1), ('(1024,512,1024) axis 0', (1024, 512, 1024), 0)):
    n = shape[axis]
    nel = shape[0] * shape[1] * shape[2]
    x = torch.randn(nel, dtype=torch.complex64, device='cuda')
    y = torch.empty_like(x)
    inner = shape[2] if axis == 1 else shape[1] * shape[2]
    dims = [(shape[0], shape[1] * shape[2], shape[1] * shape[2]), (shape[2], 1, 1)] if axis == 1 else [(shape[1], shape[2], shape[2]), (shape[2], 1, 1)]
    nbytes = nel * 8
    for _ in range(2):
        _lib.check(L.gfft_probe_copy(x.data_ptr(), y.data_ptr(), nbytes, st))
    tc = timeit(lambda: _lib.check(L.gfft_probe_copy(x.data_ptr(), y.data_ptr(), nbytes, st)))
    row = []
    want = None
    for v in (0, 10) + ((11,) if os.environ.get('GFFT_PROBE_TOUCH') else ()):       # (11: the touch experiment of round 6, no longer built)
        _lib.set_option('variant_cols', v)
        h = eng.plan_create_guru(4, -1, (n, inner, inner), dims)
        row.append(timeit(lambda: eng.execute_ptr(h, x.data_ptr(), y.data_ptr(), 1.0)))
        if v == 0:
            want = y.clone()
        elif v == 11:               # (the touches change nothing in the results)
            assert torch.equal(y, want), 'variant 11 differs from the default kernel'
        eng.plan_destroy(h)
    _lib.set_option('variant_cols', 0)
    extra = '   with touches of the next tile %.3f ms (%+.1f %%)' % (row[2], 100 * (row[2] / row[0] - 1)) if len(row) > 2 else ''
    print('%-36s kernel %.3f ms (%.0f GB/s)   pattern alone %.3f ms (%.0f GB/s)   copy %.3f ms (%.0f GB/s)%s' % (
        name, row[0], 2 * nbytes / row[0] / 1e6, row[1], 2 * nbytes / row[1] / 1e6, tc, 2 * nbytes / tc / 1e6, extra), flush=True)
    del want
    del x, y
    torch.cuda.empty_cache()
