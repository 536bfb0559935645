#!/usr/bin/env python3
"""Developer micro-benchmarks (run on the GPU box): HBM copy ceilings, the strided-tile access
pattern of column passes, and per-axis FFT passes with the selectable kernel variants.
Prints one line per measurement:  name  ms  GB/s(algorithmic: 1 read + 1 write)."""
import argparse
import ctypes
import sys
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

import numpy as np
import torch

from mpi4py_fft_amd import _lib, fftw
from mpi4py_fft_amd.array import DeviceArray


def timeit(fn, iters=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(iters):
        s.record()
        fn()
        e.record()
        e.synchronize()
        ts.append(s.elapsed_time(e))
    return min(ts), float(np.median(ts))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--n', type=int, default=1024)
    ap.add_argument('--shape', default='')
    ap.add_argument('--dtype', default='D')
    ap.add_argument('--what', default='copy,tile,fft')
    ap.add_argument('--rows', default='0,1,2,3')
    ap.add_argument('--cols', default='0,1,2,3')
    ap.add_argument('--gridcaps', default='4096')
    args = ap.parse_args()
    n = args.n
    shape = tuple(int(x) for x in args.shape.split(',')) if args.shape else (n, n, n)
    L = _lib.lib()
    print(torch.cuda.get_device_name(0), shape, flush=True)
    a = DeviceArray(shape, args.dtype)
    b = DeviceArray(shape, args.dtype)
    ar = torch.view_as_real(a.tensor)
    for i in range(0, shape[0], 64):
        ar[i:i + 64].normal_()
    nbytes = a.nbytes
    st = _lib.current_stream()

    def report(name, t):
        print('%-44s %9.3f ms (med %9.3f)  %8.1f GB/s' % (name, t[0], t[1], 2 * nbytes / t[0] / 1e6), flush=True)

    if 'copy' in args.what:
        report('copy 4x16B/thread (probe)', timeit(lambda: _lib.check(L.gfft_probe_copy(a.data_ptr, b.data_ptr, nbytes, st))))
        _lib.set_option('copy_nt', 1)
        report('copy 4x16B/thread nontemporal', timeit(lambda: _lib.check(L.gfft_probe_copy(a.data_ptr, b.data_ptr, nbytes, st))))
        _lib.set_option('copy_nt', 0)
        report('torch copy_', timeit(lambda: b.tensor.copy_(a.tensor)))
    if 'tile' in args.what and a.itemsize == 16:
        for tc in (2, 4, 8, 16, 32, 64):
            report('tile copy axis1 (stride %d el) tcols=%d' % (shape[2], tc),
                   timeit(lambda: _lib.check(L.gfft_probe_tile_copy(a.data_ptr, b.data_ptr, shape[0], shape[1], shape[2], tc, st))))
        for tc in (2, 4, 8, 16, 32, 64):
            report('tile copy axis0 (stride %d el) tcols=%d' % (shape[1] * shape[2], tc),
                   timeit(lambda: _lib.check(L.gfft_probe_tile_copy(a.data_ptr, b.data_ptr, 1, shape[0], shape[1] * shape[2], tc, st))))
    if 'fft' in args.what:
        for cap in [int(c) for c in args.gridcaps.split(',')]:
            _lib.set_option('grid_cap', cap)
            for v in [int(x) for x in args.rows.split(',')]:
                _lib.set_option('variant_rows', v)
                p = fftw.fftn(a, axes=(2,), output_array=b)
                report('fft axis2 rows variant=%d cap=%d' % (v, cap), timeit(lambda: p.execute_scaled(a, b, 1.0)))
                p.destroy()
            for v in [int(x) for x in args.cols.split(',')]:
                _lib.set_option('variant_cols', v)
                for ax in (1, 0):
                    p = fftw.fftn(a, axes=(ax,), output_array=b)
                    report('fft axis%d cols variant=%d cap=%d' % (ax, v, cap), timeit(lambda: p.execute_scaled(a, b, 1.0)))
                    p.destroy()
                p = fftw.fftn(a, axes=(1,), output_array=a)
                report('fft axis1 cols in-place variant=%d cap=%d' % (v, cap), timeit(lambda: p.execute_scaled(a, a, 1.0 / shape[1])))
                p.destroy()


if __name__ == '__main__':
    main()
