#!/usr/bin/env python3
"""Developer probe: access-pattern experiments for the strided passes (runs on the GPU box)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from mpi4py_fft_amd import _lib

L = _lib.lib()
L.gfft_debug_pass.argtypes = [ctypes.POINTER(ctypes.c_int64), ctypes.c_int, ctypes.c_int, ctypes.c_int,
                              ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]


def timeit(fn, iters=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(iters):
        s.record(); fn(); e.record(); e.synchronize()
        ts.append(s.elapsed_time(e))
    return min(ts)


def run_pass(geom, variant, a, b, cols=1):
    g = (ctypes.c_int64 * 12)(*geom)
    st = _lib.current_stream()
    return timeit(lambda: _lib.check(L.gfft_debug_pass(g, 8, cols, variant, 0, a.data_ptr(), b.data_ptr(), st)))


def main():
    n = 1024
    print(torch.cuda.get_device_name(0))
    # E1: copy bandwidth vs size
    for gib in (0.25, 1, 4, 16):
        nb = int(gib * 2 ** 30)
        x = torch.empty(nb, dtype=torch.uint8, device='cuda'); y = torch.empty_like(x)
        st = _lib.current_stream()
        t = timeit(lambda: _lib.check(L.gfft_probe_copy(x.data_ptr(), y.data_ptr(), nb, st)))
        print('copy %6.2f GiB: %8.3f ms  %7.1f GB/s' % (gib, t, 2 * nb / t / 1e6), flush=True)
        del x, y
    pad = 16
    nbytes = n * n * n * 16
    a = torch.empty(n * n * (n + pad) * 2, dtype=torch.float64, device='cuda').normal_()
    b = torch.empty_like(a)
    N2 = n * n

    def rep(name, t):
        print('%-64s %8.3f ms  %7.1f GB/s' % (name, t, 2 * nbytes / t / 1e6), flush=True)

    for v, vn in ((5, 'pattern T8'), (6, 'pattern T8 nt'), (1, 'FFT T8'), (10, 'pattern T16')):
        # natural axis-1 and axis-0 passes
        rep('%s axis1 natural' % vn, run_pass([n, n, 1, n, N2, 0, 1, n, N2, 0, 1, n], v, a, b))
        rep('%s axis0 natural' % vn, run_pass([n, 1, 1, N2, 0, 0, 1, N2, 0, 0, 1, N2], v, a, b))
        # axis-0 pass reading W[k1][i0][i2] (near stride) and writing natural out[k0][k1][i2] (far stride):
        # batch (outer=k1, inner=i2): in_os = N2, in_es = n ; out_os = n, out_es = N2
        rep('%s axis0: read near (W[k1][i0][i2]) write far' % vn, run_pass([n, n, 1, n, N2, 0, 1, n, n, 0, 1, N2], v, a, b))
        # axis-1 pass reading natural (near) and writing W[k1][i0][i2] (far)
        rep('%s axis1: read near, write far (W[k1][i0][i2])' % vn, run_pass([n, n, 1, n, N2, 0, 1, n, n, 0, 1, N2], v, a, b))
        # read far / write near
        rep('%s read far write near' % vn, run_pass([n, n, 1, n, n, 0, 1, N2, N2, 0, 1, n], v, a, b))
        # padded pitch on the contiguous axis (intermediate buffers only): pitch n+pad
        P = n + pad
        rep('%s axis1 padded pitch in+out' % vn, run_pass([n, n, 1, n, n * P, 0, 1, P, n * P, 0, 1, P], v, a, b))
        rep('%s axis0 padded pitch in+out' % vn, run_pass([n, n, 1, n, P, 0, 1, n * P, P, 0, 1, n * P], v, a, b))
        rep('%s axis0 padded in, natural out' % vn, run_pass([n, n, 1, n, P, 0, 1, n * P, n, 0, 1, N2], v, a, b))
        rep('%s axis1 natural in, padded out' % vn, run_pass([n, n, 1, n, N2, 0, 1, n, n * P, 0, 1, P], v, a, b))


if __name__ == '__main__':
    main()
