#!/usr/bin/env python3
"""Developer probe (round 3): the serial stages of one rank of the pipelined C4 / C5 transforms on 8
GPUs, timed on one GPU -- the stage plans exactly as pipeline.Pipeline builds them, on the line-aligned
exchange buffers (pipeline._Aligned) and on the C-order ones of round 2.

  python tools/stage_probe.py [c4|c5|c5odd|all]     (c5odd = a rank of the 513-wide grid column)
"""
import os, sys
from types import SimpleNamespace as NS
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from mpi4py_fft_amd import _lib
from mpi4py_fft_amd import pipeline as P


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


def fake_stage(axis, shape_in, shape_out=None):
    return NS(axes=(axis,), _padded=False, forward=NS(input_array=NS(shape=shape_in), output_array=NS(shape=shape_out or shape_in)))


def run_case(name, prec, real0, sh0, nh, sh1, sh2, p0, K0, widths, p1, K1):
    isz = 2 * prec
    stages = [fake_stage(2, sh0, sh0[:2] + (nh,)), fake_stage(1, sh1), fake_stage(0, sh2)]
    eng = _lib.engine()
    nbuf = int(max(np.prod(sh1), np.prod(sh2), np.prod(sh0[:2]) * nh) * 1.25) + (1 << 20)
    cdt = torch.complex64 if prec == 4 else torch.complex128
    a = torch.randn(nbuf, dtype=cdt, device='cuda')
    b = torch.empty_like(a)
    alg = [np.prod(sh0) * (prec if real0 else isz) + np.prod(sh0[:2]) * nh * isz, 2 * np.prod(sh1) * isz, 2 * np.prod(sh2) * isz]

    def report(layout, fwd, bwd):
        for i in range(3):
            for tag, st in (('fwd', fwd[i]), ('bwd', bwd[i])):
                if st is None:
                    continue
                t = timeit(lambda: [st.execute(eng, c, a.data_ptr(), b.data_ptr(), 1.0) for c in range(st.nchunks)])
                print('%-16s %-9s stage %d %s  %d step(s) %8.3f ms  %7.1f GB/s  %4.1f %%' % (
                    name, layout, i, tag, st.nchunks, t, alg[i] / t / 1e6, alg[i] / t / 1e6 / 80), flush=True)

    pipe = NS(precision=prec, isz=isz)
    plan = [dict(p=p0, K=K0, uneven=real0, widths=widths if real0 else None), dict(p=p1, K=K1, uneven=False)]
    al = P._Aligned(pipe, stages, plan, real0)
    assert al.ok, name
    only = os.environ.get('STAGE_PROBE_ONLY')           # profiling runs: one layout per process
    if only in (None, 'aligned'):
        report('aligned', al.fwd, al.bwd)
    # round 2's C-order exchange buffers
    lin = [P.Layout(sh0), P.Layout(sh1, 1, p0, 0, K0), P.Layout(sh2, 0, p1, 2, K1)]
    lout = [P.Layout(sh0, 2, p0, 0, K0) if not real0 else P.Layout(sh0), P.Layout(sh1, 1, p1, 2, K1), P.Layout(sh2)]
    fwd = [None if real0 else P._Stage(sh0, 2, lin[0], lout[0], -1, prec), P._Stage(sh1, 1, lin[1], lout[1], -1, prec),
           P._Stage(sh2, 0, lin[2], lout[2], -1, prec)]
    bwd = [None if real0 else P._Stage(sh0, 2, lout[0], lin[0], +1, prec), P._Stage(sh1, 1, lout[1], lin[1], +1, prec),
           P._Stage(sh2, 0, lout[2], lin[2], +1, prec)]
    if real0:
        fwd[0], bwd[0] = P._RealRows(sh0, p0, K0, True, prec), P._RealRows(sh0, p0, K0, False, prec)
    if only in (None, 'c-order'):
        report('c-order', fwd, bwd)
    for s in al.fwd + al.bwd + fwd + bwd:
        if s is not None:
            s.destroy()
    del a, b
    torch.cuda.empty_cache()


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else 'all'
    print(torch.cuda.get_device_name(0))
    if what == 'chunks':
        for K0, K1 in ((1, 1), (2, 2), (4, 1), (1, 4)):
            run_case('C4@8 K0=%d K1=%d' % (K0, K1), 8, False, (256, 512, 1024), 1024, (256, 1024, 512), (1024, 256, 512), 2, K0, None, 4, K1)
        for K0 in (1, 2):
            run_case('C5@8e K0=%d' % K0, 4, True, (512, 1024, 2048), 1025, (512, 2048, 512), (2048, 512, 512), 2, K0, [513, 512], 4, 1)
        return
    if what in ('c4', 'all'):
        run_case('C4@8', 8, False, (256, 512, 1024), 1024, (256, 1024, 512), (1024, 256, 512), 2, 4, None, 4, 4)
    if what in ('c5', 'all'):
        run_case('C5@8e', 4, True, (512, 1024, 2048), 1025, (512, 2048, 512), (2048, 512, 512), 2, 4, [513, 512], 4, 1)
    if what in ('c5odd', 'all'):
        run_case('C5@8o', 4, True, (512, 1024, 2048), 1025, (512, 2048, 513), (2048, 512, 513), 2, 4, [513, 512], 4, 1)


if __name__ == '__main__':
    main()
