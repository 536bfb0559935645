#!/usr/bin/env python3
"""Developer probe (round 3): what layout of the INTERNAL stage arrays / exchange buffers of a
multi-GPU transform the strided stages run fastest on.  One power-of-two pass per case through
gfft_debug_pass (explicit strides on both sides), full tiles only:

  near      [o][n][inner]            the natural middle-axis layout (element stride = row width)
  far       [n][m][inner]            the natural outermost-axis layout (element stride = slab)
  far+skew  far with the slab stride 256 B off the power of two
  tiles     [o][tile][n][T]          tile-major: a tile's T adjacent columns lie back to back for
                                     consecutive entries of the transformed axis (one contiguous run)
Shapes: config C5 on 8 GPUs (c64, n = 2048) and config C4 on 8 GPUs (c128, n = 1024)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mpi4py_fft_amd import _lib

L = _lib.lib()
L.gfft_debug_pass.argtypes = [ctypes.POINTER(ctypes.c_int64), ctypes.c_int, ctypes.c_int, ctypes.c_int,
                              ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]


def timeit(fn, iters=7, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(iters):
        s.record(); fn(); e.record(); e.synchronize()
        ts.append(s.elapsed_time(e))
    return min(ts)


def run(prec, geom, a, b, variant=0):
    g = (ctypes.c_int64 * 12)(*geom)
    st = _lib.current_stream()
    return timeit(lambda: _lib.check(L.gfft_debug_pass(g, prec, 1, variant, 0, a.data_ptr(), b.data_ptr(), st)))


def side(kind, n, o, w, T=16, skew=0):
    """(outer, mid, inner, os, ms, is, es) of one side for `o` slabs of `w` adjacent columns."""
    if kind == 'near':            # [o][n][w]
        return dict(os=n * w, ms=T, es=w)
    if kind == 'far':             # [n][o][w] (+ skew elements per slab of the transformed axis)
        return dict(os=w, ms=T, es=o * w + skew)
    if kind == 'tiles':           # [o][w/T][n][T]
        return dict(os=n * w, ms=n * T, es=T)
    if kind == 'tiles_far':       # [w/T][n][o][T]: tile-major whose runs are o*T long per entry  (a far scatter)
        return dict(os=T, ms=n * o * T, es=o * T)
    raise ValueError(kind)


def case(prec, n, o, w, kin, kout, a, b, T=16, skew_in=0, skew_out=0, variant=0):
    si, so = side(kin, n, o, w, T, skew_in), side(kout, n, o, w, T, skew_out)
    geom = [n, o, w // T, T, si['os'], si['ms'], 1, si['es'], so['os'], so['ms'], 1, so['es']]
    t = run(prec, geom, a, b, variant)
    nbytes = 2.0 * n * o * w * 2 * prec
    print('  n=%d %s (%d x %d cols)  %-10s -> %-10s %s %8.3f ms  %7.1f GB/s  %4.1f %%' % (
        n, 'c64' if prec == 4 else 'c128', o, w, kin + ('+%d' % skew_in if skew_in else ''),
        kout + ('+%d' % skew_out if skew_out else ''), ('v%d' % variant) if variant else '  ', t, nbytes / t / 1e6,
        nbytes / t / 1e6 / 80), flush=True)


def wide_tiles(prec, n, o, w, W, a, b, kout='near'):
    """Input in layout tiles of W > T columns ([o][w/W][n][W]) read by the kernel's T = 16 column tiles:
    a kernel tile takes a T-wide slice of every W-wide row of its layout tile."""
    T = 16
    so = side(kout, n, o, w, T)
    # in: (o, w/W) merge into one outer dim of stride n*W; mid = W/T slices (stride T); es = W
    if kout == 'near':
        # out natural [o][n][w]: outer (o, kk) does not merge on this side unless w == W * (w/W): os = W per kk, n*w per o
        # -> run one launch per o-slab group is overkill for a probe: use out = same wide-tile layout
        pass
    geom = [n, o * (w // W), W // T, T, n * W, T, 1, W, n * W, T, 1, W]
    t = run(prec, geom, a, b)
    nbytes = 2.0 * n * o * w * 2 * prec
    print('  n=%d %s (%d x %d cols)  tiles%-5d -> tiles%-5d    %8.3f ms  %7.1f GB/s  %4.1f %%' % (
        n, 'c64' if prec == 4 else 'c128', o, w, W, W, t, nbytes / t / 1e6, nbytes / t / 1e6 / 80), flush=True)


def main():
    print(torch.cuda.get_device_name(0))
    if os.environ.get('PROBE_WIDE'):
        prec, n, o, w = 4, 2048, 512, 512
        a = torch.randn(n * o * w + (1 << 22), dtype=torch.complex64, device='cuda')
        b = torch.empty_like(a)
        case(prec, n, o, w, 'near', 'near', a, b)
        case(prec, n, o, w, 'tiles', 'tiles', a, b)
        for W in (32, 64):
            wide_tiles(prec, n, o, w, W, a, b)
        return
    kinds = ['near', 'far', 'tiles', 'tiles_far']
    for prec, n, o, w, dt in ((4, 2048, 512, 512, torch.complex64), (8, 1024, 256, 512, torch.complex128)):
        a = torch.randn(n * o * w + (1 << 22), dtype=dt, device='cuda')
        b = torch.empty_like(a)
        sk = 256 // (2 * prec)
        print('--- %s n=%d: every in/out combination' % ('c64' if prec == 4 else 'c128', n))
        for ki in kinds:
            for ko in kinds:
                case(prec, n, o, w, ki, ko, a, b)
        print('--- skewed far sides')
        case(prec, n, o, w, 'far', 'far', a, b, skew_in=sk, skew_out=sk)
        case(prec, n, o, w, 'far', 'far', a, b, skew_in=2 * sk, skew_out=2 * sk)
        case(prec, n, o, w, 'far', 'far', a, b, skew_in=o * 1, skew_out=o * 1)      # = rows one element wider (513-style)
        case(prec, n, o, w, 'far', 'near', a, b, skew_in=sk)
        case(prec, n, o, w, 'far', 'tiles', a, b, skew_in=sk)
        case(prec, n, o, w, 'near', 'far', a, b, skew_out=sk)
        case(prec, n, o, w, 'tiles', 'far', a, b, skew_out=sk)
        if prec == 4:
            print('--- kernel variants on near -> near and tiles -> tiles')
            for v in (1,):
                case(prec, n, o, w, 'near', 'near', a, b, variant=v)
                case(prec, n, o, w, 'tiles', 'tiles', a, b, variant=v)
        del a, b
        torch.cuda.empty_cache()
    # C4 on 8 GPUs, middle stage (256, 1024, 512) c128 axis 1
    prec, n, o, w = 8, 1024, 256, 512
    a = torch.randn(n * o * w + (1 << 22), dtype=torch.complex128, device='cuda')
    b = torch.empty_like(a)
    print('--- c128 n=1024 T=16 is the kernel tile; layout tiles of 16')
    for ki, ko in (('near', 'near'), ('tiles', 'near'), ('near', 'tiles'), ('tiles', 'tiles')):
        case(prec, n, o, w, ki, ko, a, b)


if __name__ == '__main__':
    main()
