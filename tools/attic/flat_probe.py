#!/usr/bin/env python3
"""Developer probe: the pass that writes a 3-D r2c transform's 513-wide output.  Now: along axis 1
(near), tiles inside a row -> misaligned stores.  Candidate: along axis 0 (far) with tiles over the
flattened (i1, c) index -> aligned stores, loads from the pitched workspace misaligned instead."""
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


def run_pass(geom, a, b, flat, swz):
    _lib.set_option('debug_flat', flat)
    _lib.set_option('xcd_swizzle', swz)
    g = (ctypes.c_int64 * 12)(*geom)
    st = _lib.current_stream()
    t = timeit(lambda: _lib.check(L.gfft_debug_pass(g, 8, 1, 0, 0, a.data_ptr(), b.data_ptr(), st)))
    _lib.set_option('debug_flat', 0)
    _lib.set_option('xcd_swizzle', -1)
    return t


n, nc, P = 1024, 513, 520
print(torch.cuda.get_device_name(0))
a = torch.empty((n * (n * P + 16) + 64) * 2, dtype=torch.float64, device='cuda').normal_()
b = torch.empty(n * n * nc * 2 + 64, dtype=torch.float64, device='cuda')
gb = n * n * nc * 32 / 1e6
for swz in (0, 1):
    # now: W[i1][i0][c] -> OUT[i0][i1][c], transform along i1: batch (o = i0, i = c)
    t = run_pass([n, n, 1, P, P, 0, 1, n * P, n * nc, 0, 1, nc], a, b, 0, swz)
    print('axis 1 (near), row tiles, W[i1][i0][c] -> natural   swizzle %d  %7.3f ms %7.1f GB/s' % (swz, t, gb / t), flush=True)
    # candidate: W[i0][i1][c] (plane pitch n*P + 16) -> OUT, transform along i0: flattened tiles
    PL = n * P + 16
    t = run_pass([n, 1, n, nc, 0, P, 1, PL, 0, nc, 1, n * nc], a, b, 1, swz)
    print('axis 0 (far), flat tiles, W[i0][i1][c] -> natural   swizzle %d  %7.3f ms %7.1f GB/s' % (swz, t, gb / t), flush=True)
    PL = n * P
    t = run_pass([n, 1, n, nc, 0, P, 1, PL, 0, nc, 1, n * nc], a, b, 1, swz)
    print('   ... plane pitch n*P (no extra 256 B)              swizzle %d  %7.3f ms %7.1f GB/s' % (swz, t, gb / t), flush=True)
    # reference: the aligned in-workspace pass
    t = run_pass([n, n, 1, P, n * P, 0, 1, P, n * P, 0, 1, P], a, a, 0, swz)
    print('in-workspace pass (aligned both sides, in place)     swizzle %d  %7.3f ms %7.1f GB/s' % (swz, t, gb / t), flush=True)
print('backward direction (first pass reads the natural 513-wide array)')
for swz in (0, 1):
    t = run_pass([n, n, 1, P, n * nc, 0, 1, nc, P, 0, 1, n * P], b, a, 0, swz)
    print('axis 1 (near), row tiles, natural -> W[i1][i0][c]   swizzle %d  %7.3f ms %7.1f GB/s' % (swz, t, gb / t), flush=True)
    PL = n * P + 16
    t = run_pass([n, 1, n, nc, 0, nc, 1, n * nc, 0, P, 1, PL], b, a, 1, swz)
    print('axis 0 (far), flat tiles, natural -> W[i0][i1][c]   swizzle %d  %7.3f ms %7.1f GB/s' % (swz, t, gb / t), flush=True)
