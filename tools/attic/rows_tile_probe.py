#!/usr/bin/env python3
"""Developer probe (round 3): what a ROW pass pays for writing / reading a tile-major exchange buffer
([slab][tile][row][16] instead of [slab][row][n]) -- the layout that makes the NEXT stage's strided
pass one contiguous run per tile (tools/stage_layout_probe.py)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mpi4py_fft_amd import _lib

L = _lib.lib()
L.gfft_debug_pass.argtypes = [ctypes.POINTER(ctypes.c_int64), ctypes.c_int, ctypes.c_int, ctypes.c_int,
                              ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]


def timeit(fn, iters=9, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(iters):
        s.record(); fn(); e.record(); e.synchronize()
        ts.append(s.elapsed_time(e))
    return min(ts)


def case(prec, n, n0, n1, side, a, b, lg=4):
    # rows (i0, i1): natural [i0][i1][n]; tile-major [i0][n >> lg][i1][1 << lg]
    T = 1 << lg
    nat = dict(os=n1 * n, is_=n)
    til = dict(os=n1 * n, is_=T)
    si = til if side & 1 else nat
    so = til if side & 2 else nat
    geom = [n, n0, 1, n1, si['os'], 0, si['is_'], 1, so['os'], 0, so['is_'], 1]
    _lib.set_option('debug_tile_lg', lg)
    _lib.set_option('debug_tile_side', side)
    _lib.set_option('debug_tile_stride', n1 * T)
    g = (ctypes.c_int64 * 12)(*geom)
    st = _lib.current_stream()
    t = timeit(lambda: _lib.check(L.gfft_debug_pass(g, prec, 0, 0, 0, a.data_ptr(), b.data_ptr(), st)))
    _lib.set_option('debug_tile_side', 0)
    nbytes = 2.0 * n * n0 * n1 * 2 * prec
    print('  rows n=%d %s (%d x %d rows) %-8s -> %-8s %8.3f ms  %7.1f GB/s  %4.1f %%' % (
        n, 'c64' if prec == 4 else 'c128', n0, n1, 'tiles' if side & 1 else 'natural', 'tiles' if side & 2 else 'natural',
        t, nbytes / t / 1e6, nbytes / t / 1e6 / 80), flush=True)


def main():
    print(torch.cuda.get_device_name(0))
    for prec, n, n0, n1, dt in ((8, 1024, 256, 512, torch.complex128), (4, 1024, 512, 1024, torch.complex64),
                                (4, 2048, 512, 512, torch.complex64), (8, 1024, 64, 512, torch.complex128)):
        a = torch.randn(n * n0 * n1, dtype=dt, device='cuda')
        b = torch.empty_like(a)
        for side in (0, 2, 1, 3):
            case(prec, n, n0, n1, side, a, b)
        if prec == 4:
            for lg in (5, 6):
                print('  tiles of %d elements:' % (1 << lg))
                for side in (2, 1):
                    case(prec, n, n0, n1, side, a, b, lg)
        del a, b
        torch.cuda.empty_cache()


if __name__ == '__main__':
    main()
