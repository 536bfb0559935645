#!/usr/bin/env python3
"""Developer probe (round 6): why the backward per-rank stages of C4 on 8 GPUs run 10-17 % slower than their forward
twins (profiles/r05_stage_probe.txt: stage 1 0.811 / 0.897 ms, stage 2 0.908 / 1.060 ms).  Separates the DIRECTION
(conjugation on load / store: same kernel, a sign) from the LAYOUT each direction reads and writes, on one strided
pass at the stage shapes: natural <-> natural, pitched <-> natural, pitched <-> pitched, under the strided-kernel
variants (option variant_cols: 0 = 32 values per thread + non-temporal streams, 15 = the same with plain loads and
stores, 17 = 16 values per thread on 1024 threads), tile orders and grid caps.

  python tools/bwd_probe.py [stage2|stage1|all]
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mpi4py_fft_amd import _lib
from mpi4py_fft_amd import pipeline as P


def timeit(fn, iters=15, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(iters):
        s.record(); fn(); e.record(); e.synchronize()
        ts.append(s.elapsed_time(e))
    ts.sort()
    return ts[0], ts[len(ts) // 2]


def probe(name, n, outer, inner, far):
    """One strided pass of length n over an array [outer][n][inner] (far = False) or [n][outer][inner] (far = True:
    the transformed axis is the outermost one); 'pitched' = the slabs along the transformed axis 256 bytes off the
    power of two."""
    eng = _lib.engine()
    isz = 16
    if far:
        es_nat = outer * inner
        es_pit = P._pitch(es_nat, isz)
        dims = lambda a, b: [(outer, inner, inner), (inner, 1, 1)]
        size = n * es_pit
    else:
        es_nat = inner
        es_pit = inner            # (near axis: rows stay where they are; the planes get the pitch)
        pl_nat, pl_pit = n * inner, P._pitch(n * inner, isz)
        size = outer * pl_pit
    x = torch.randn(size, dtype=torch.complex128, device='cuda')
    y = torch.empty(size, dtype=torch.complex128, device='cuda')
    alg = 2.0 * outer * n * inner * isz
    for opts in (dict(), dict(variant_cols=15), dict(variant_cols=17), dict(xcd_swizzle=1), dict(grid_cap=1024), dict(grid_cap=16384)):
        for k in ('variant_cols', 'xcd_swizzle', 'grid_cap'):
            _lib.set_option(k, {'variant_cols': 0, 'xcd_swizzle': -1, 'grid_cap': 0}[k])
        for k, v in opts.items():
            _lib.set_option(k, v)
        for lin in ('nat', 'pit'):
            for lout in ('nat', 'pit'):
                row = []
                for kind in (-1, +1):
                    if far:
                        si, so = (es_nat if lin == 'nat' else es_pit), (es_nat if lout == 'nat' else es_pit)
                        h = eng.plan_create_guru(8, kind, (n, si, so), [(outer, inner, inner), (inner, 1, 1)])
                    else:
                        pi, po = (pl_nat if lin == 'nat' else pl_pit), (pl_nat if lout == 'nat' else pl_pit)
                        h = eng.plan_create_guru(8, kind, (n, inner, inner), [(outer, pi, po), (inner, 1, 1)])
                    t = timeit(lambda: eng.execute_ptr(h, x.data_ptr(), y.data_ptr(), 1.0))
                    row.append(t)
                    eng.plan_destroy(h)
                print('%-8s %-22s %s -> %s   fwd %.3f (med %.3f)  bwd %.3f (med %.3f) ms   %4.1f / %4.1f %%' % (
                    name, opts or 'default', lin, lout, row[0][0], row[0][1], row[1][0], row[1][1],
                    alg / row[0][0] / 8e7, alg / row[1][0] / 8e7), flush=True)
    for k in ('variant_cols', 'grid_cap'):
        _lib.set_option(k, 0)
    _lib.set_option('xcd_swizzle', -1)
    del x, y
    torch.cuda.empty_cache()


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else 'all'
    print(torch.cuda.get_device_name(0))
    if what in ('stage2', 'all'):
        probe('stage2', 1024, 256, 512, True)          # (1024, 256, 512) along axis 0
    if what in ('stage1', 'all'):
        probe('stage1', 1024, 256, 512, False)         # (256, 1024, 512) along axis 1


if __name__ == '__main__':
    main()
