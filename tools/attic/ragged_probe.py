#!/usr/bin/env python3
"""Developer probe: the in-workspace strided pass of an r2c 3-D plan (rows of pitch 520, 513 used)
against the same geometry with 512 / 520 columns used, in place and out of place."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mpi4py_fft_amd import _lib

L = _lib.lib()


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


def run_pass(geom, a, b, variant=0):
    g = (ctypes.c_int64 * 12)(*geom)
    st = _lib.current_stream()
    return timeit(lambda: _lib.check(L.gfft_debug_pass(g, 8, 1, variant, 0, a.data_ptr(), b.data_ptr(), st)))


n = 1024
print(torch.cuda.get_device_name(0))
for P in (520, 528, 576):
    a = torch.empty(n * n * P * 2, dtype=torch.float64, device='cuda').normal_()
    b = torch.empty_like(a)
    for used in (512, 513, 520):
        for inplace in (False, True):
            # W[i1][i0][c]: batch (o = i1, i = c), es = P, os = n * P
            t = run_pass([n, n, 1, used, n * P, 0, 1, P, n * P, 0, 1, P], a, a if inplace else b)
            print('near axis in W pitch %d used %d %-9s %7.3f ms %7.1f GB/s' % (P, used, 'in place' if inplace else 'out', t, n * n * used * 32 / t / 1e6), flush=True)
            # far axis: batch (o = i0 ... ) es = n*P, os = P
            t = run_pass([n, n, 1, used, P, 0, 1, n * P, P, 0, 1, n * P], a, a if inplace else b)
            print('far  axis in W pitch %d used %d %-9s %7.3f ms %7.1f GB/s' % (P, used, 'in place' if inplace else 'out', t, n * n * used * 32 / t / 1e6), flush=True)
    del a, b
