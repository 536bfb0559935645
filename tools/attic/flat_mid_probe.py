#!/usr/bin/env python3
"""Developer probe: middle-axis strided passes over rows of odd width (257-, 513-wide half spectra).
Row tiles (T columns inside one row: the last tile of every row holds ONE column) against tiles over
the flattened (i0, c) index (no ragged tiles; the outer axis is passed as PassDesc::mid)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mpi4py_fft_amd import _lib

L = _lib.lib()


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


def run_pass(geom, prec, a, b, flat, swz):
    _lib.set_option('debug_flat', flat)
    _lib.set_option('xcd_swizzle', swz)
    g = (ctypes.c_int64 * 12)(*geom)
    st = _lib.current_stream()
    t = timeit(lambda: _lib.check(L.gfft_debug_pass(g, prec, 1, 0, 0, a.data_ptr(), b.data_ptr(), st)))
    _lib.set_option('debug_flat', 0)
    _lib.set_option('xcd_swizzle', -1)
    return t


print(torch.cuda.get_device_name(0))
for (n0, n1, w, prec) in ((1536, 1024, 257, 4), (2048, 512, 257, 4), (512, 2048, 513, 4), (1024, 1024, 513, 8), (512, 2048, 512, 4), (256, 1024, 513, 8)):
    dt = torch.float32 if prec == 4 else torch.float64
    a = torch.empty(n0 * n1 * w * 2 + 64, dtype=dt, device='cuda').normal_()
    b = torch.empty(n0 * n1 * w * 2 + 64, dtype=dt, device='cuda')
    gb = n0 * n1 * w * 4 * prec / 1e6
    res = []
    for swz in (0, 1):
        t0 = run_pass([n1, n0, 1, w, n1 * w, 0, 1, w, n1 * w, 0, 1, w], prec, a, b, 0, swz)
        t1 = run_pass([n1, 1, n0, w, 0, n1 * w, 1, w, 0, n1 * w, 1, w], prec, a, b, 1, swz)
        res.append('swizzle %d: row tiles %6.3f ms %6.0f GB/s | flat tiles %6.3f ms %6.0f GB/s' % (swz, t0, gb / t0, t1, gb / t1))
    # same results?
    _lib.set_option('debug_flat', 0)
    run_pass([n1, n0, 1, w, n1 * w, 0, 1, w, n1 * w, 0, 1, w], prec, a, b, 0, 0)
    ref = b.clone()
    run_pass([n1, 1, n0, w, 0, n1 * w, 1, w, 0, n1 * w, 1, w], prec, a, b, 1, 0)
    same = torch.equal(ref, b)
    print('(%d,%d,%d) %s axis 1  same=%s' % (n0, n1, w, 'c64' if prec == 4 else 'c128', same))
    for r in res:
        print('   ' + r, flush=True)
    del a, b, ref
