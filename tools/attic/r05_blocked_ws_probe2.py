#!/usr/bin/env python3
"""Developer probe (round 5): the strided pass between a NATURAL array and the pitched workspace of the 3-D schedules, lines over
axis 1, three workspace layouts: W[k1][i0][c] (rows of a line n P apart: what the unfused schedules do), W[i0][k1][c] (P apart:
the pair schedule) and the BLOCKED W[k1/16][i0][k1%16][c] (16 rows P apart, then a jump: tile-major LINES of a strided pass,
PassDesc::in_tlg / out_tlg).  Both directions (natural -> W, W -> natural); the blocked results checked against the plain ones."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mpi4py_fft_amd import _lib

L = _lib.lib()
B = 16


def timeit(fn, iters=6, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); e.synchronize()
        ts.append(s.elapsed_time(e))
    return min(ts)


def run(geom, prec, a, b, side=0, stride=0, time=True):
    _lib.set_option('debug_tile_side', side)
    _lib.set_option('debug_tile_lg', 4)
    _lib.set_option('debug_tile_stride', stride)
    g = (ctypes.c_int64 * 12)(*geom)
    st = _lib.current_stream()
    fn = lambda: _lib.check(L.gfft_debug_pass(g, prec, 1, 0, 0, a.data_ptr(), b.data_ptr(), st))
    t = timeit(fn) if time else fn()
    _lib.set_option('debug_tile_side', 0)
    return t


print(torch.cuda.get_device_name(0))
for n, prec, P in ((1024, 8, 1040), (512, 8, 528), (1024, 4, 1056), (256, 8, 272)):
    dt = torch.complex128 if prec == 8 else torch.complex64
    nat = torch.randn(n * n * n, dtype=dt, device='cuda')
    w = torch.zeros(n * n * P, dtype=dt, device='cuda')
    w2 = torch.zeros(n * n * P, dtype=dt, device='cuda')
    gb = 2 * n ** 3 * 2 * prec / 1e6
    assert n * B * P < 2 ** 31
    tag = 'n = %d %s' % (n, 'c128' if prec == 8 else 'c64')
    # natural [i0][i1][c] -> W, lines over i1: geom = n, outer (i0), mid, inner (c), in_os, in_ms, in_is, in_es, out_os, out_ms, out_is, out_es
    far = [n, n, 1, n, n * n, 0, 1, n, P, 0, 1, n * P]
    near = [n, n, 1, n, n * n, 0, 1, n, n * P, 0, 1, P]
    blk = [n, n, 1, n, n * n, 0, 1, n, B * P, 0, 1, P]
    for rnd in range(2):
        print('%s  natural -> W[k1][i0][c]         %.3f ms  %.0f GB/s' % (tag, run(far, prec, nat, w), gb / run(far, prec, nat, w)), flush=True)
        print('%s  natural -> W[i0][k1][c]         %.3f ms  %.0f GB/s' % (tag, run(near, prec, nat, w2), gb / run(near, prec, nat, w2)), flush=True)
        t = run(blk, prec, nat, w2, 2, n * B * P)
        print('%s  natural -> W[k1/16][i0][k1%%16][c] %.3f ms  %.0f GB/s' % (tag, t, gb / t), flush=True)
    # the blocked result against the plain one
    run(far, prec, nat, w, time=False); run(blk, prec, nat, w2, 2, n * B * P, time=False)
    torch.cuda.synchronize()
    A = w.view(n, n, P)[:, :, :n]                                              # [k1][i0][c]
    Bk = w2.view(n // B, n, B, P)[:, :, :, :n].permute(0, 2, 1, 3).reshape(n, n, n)    # [k1/16][i0][k1%16][c] -> [k1][i0][c]
    print('%s  blocked store == plain store: %s' % (tag, bool(torch.equal(A, Bk))), flush=True)
    # W -> natural, lines over i1 (the forward direction's last pass)
    farr = [n, n, 1, n, P, 0, 1, n * P, n * n, 0, 1, n]
    nearr = [n, n, 1, n, n * P, 0, 1, P, n * n, 0, 1, n]
    blkr = [n, n, 1, n, B * P, 0, 1, P, n * n, 0, 1, n]
    out1, out2 = torch.empty_like(nat), torch.empty_like(nat)
    for rnd in range(2):
        print('%s  W[k1][i0][c] -> natural         %.3f ms' % (tag, run(farr, prec, w, out1)), flush=True)
        print('%s  W[i0][k1][c] -> natural         %.3f ms' % (tag, run(nearr, prec, w, out1)), flush=True)
        print('%s  W[k1/16][i0][k1%%16][c] -> natural %.3f ms' % (tag, run(blkr, prec, w2, out2, 1, n * B * P)), flush=True)
    run(farr, prec, w, out1, time=False); run(blkr, prec, w2, out2, 1, n * B * P, time=False)
    torch.cuda.synchronize()
    print('%s  blocked load == plain load: %s' % (tag, bool(torch.equal(out1, out2))), flush=True)
    del nat, w, w2, out1, out2
    torch.cuda.empty_cache()
