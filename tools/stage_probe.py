#!/usr/bin/env python3
"""Developer probe (round 3): the serial stages of one rank of the pipelined C4 / C5 transforms on 8
GPUs, timed on one GPU -- the stage plans exactly as pipeline.Pipeline builds them, on the line-aligned
exchange buffers (pipeline._Aligned) and on the C-order ones of round 2.

  python tools/stage_probe.py [slab|c4|c5|c5odd|all]     (c5odd = a rank of the 513-wide grid column; slab = C3, C4 on 2 ranks
  and C4 on the (8,1,1) grid: the two local stages as two launches against one fused launch, round 6)
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

    only_dir, only_stage = os.environ.get('STAGE_PROBE_DIR'), os.environ.get('STAGE_PROBE_STAGE')     # profiling runs: one direction / stage per process

    def report(layout, fwd, bwd):
        for i in range(3):
            for tag, st in (('fwd', fwd[i]), ('bwd', bwd[i])):
                if st is None or (only_dir and tag != only_dir) or (only_stage and str(i) != only_stage):
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


def run_slab(name, prec, N0, N1, N2, p, Ks=(1, 2, 4)):
    """The per-rank stages of a complex transform on a SLAB grid (first redistribution local): the two local stages as
    two stand-alone plans on the shared buffer (round 5: rows natural, then strided into the packed send buffer), as
    one fused launch (gfft_plan_create_guru2) whole and per chunk of planes, and the far stage on pitched slabs."""
    isz = 2 * prec
    eng = _lib.engine()
    cdt = torch.complex64 if prec == 4 else torch.complex128
    M0, N1b = p * N0, N1 // p
    E = P._pitch(N1b * N2, isz)
    nel = N0 * N1 * N2
    a = torch.randn(nel, dtype=cdt, device='cuda')
    mid = torch.empty(nel, dtype=cdt, device='cuda')
    b = torch.empty(max(p * N0 * E, M0 * E), dtype=cdt, device='cuda')
    c = torch.empty(nel, dtype=cdt, device='cuda')
    alg2, alg1 = 4 * nel * isz, 2 * nel * isz

    def line(tag, t, alg, extra=''):
        print('%-12s %-34s %8.3f ms  %7.1f GB/s  %4.1f %%  %s' % (name, tag, t, alg / t / 1e6, alg / t / 1e6 / 80, extra), flush=True)
    # two stand-alone launches: rows natural -> natural, strided natural -> packed [block][plane][rows of the block][cols]
    for kind, tag in ((-1, 'fwd'), (+1, 'bwd')):
        hr = eng.plan_create_guru(prec, kind, (N2, 1, 1), [(N0 * N1, N2, N2)], 1, 0, 1, 0)
        if kind < 0:
            hc = eng.plan_create_guru(prec, kind, (N1, N2, N2), [(N0, N1 * N2, N1b * N2), (N2, 1, 1)], 1, 0, p, N0 * N1b * N2)
            t = timeit(lambda: (eng.execute_ptr(hr, a.data_ptr(), mid.data_ptr(), 1.0), eng.execute_ptr(hc, mid.data_ptr(), b.data_ptr(), 1.0)))
        else:
            hc = eng.plan_create_guru(prec, kind, (N1, N2, N2), [(N0, N1b * N2, N1 * N2), (N2, 1, 1)], p, N0 * N1b * N2, 1, 0)
            t = timeit(lambda: (eng.execute_ptr(hc, b.data_ptr(), mid.data_ptr(), 1.0), eng.execute_ptr(hr, mid.data_ptr(), a.data_ptr(), 1.0)))
        line('two launches %s' % tag, t, alg2)
        eng.plan_destroy(hr)
        eng.plan_destroy(hc)
    for K in Ks:
        if N0 % K:
            continue
        for fwd in (True, False):
            st = P._PairStage((N0, N1, N2), p, K, E, fwd, prec)
            if st.plan is None:
                continue
            pin, pout = (a, b) if fwd else (b, c)
            t = timeit(lambda: [st.execute(eng, q, pin.data_ptr(), pout.data_ptr(), 1.0) for q in range(K)])
            line('pair K=%d %s (%d launch%s per chunk)' % (K, 'fwd' if fwd else 'bwd', st.launches, '' if st.launches == 1 else 'es'), t, alg2)
            st.destroy()
    # (the forward pair the other way round: rows first, the strided pass storing into the blocks)
    h = eng.plan_create_guru2(prec, -1, (N1, N2, N2), (N2, 1, 1), (N0, N1 * N2, E), False, 1, 0, p, N0 * E)
    if h is not None:
        line('pair K=1 fwd [rows -> strided] (%d)' % eng.plan_cost(h)[2], timeit(lambda: eng.execute_ptr(h, a.data_ptr(), b.data_ptr(), 1.0)), alg2)
        eng.plan_destroy(h)
    # the far stage: axis 0 over slabs E apart <-> the natural output
    x = torch.randn(M0 * E, dtype=cdt, device='cuda')
    y = torch.empty(M0 * N1b * N2, dtype=cdt, device='cuda')
    for kind, tag in ((-1, 'fwd'), (+1, 'bwd')):
        si, so = (E, N1b * N2) if kind < 0 else (N1b * N2, E)
        h = eng.plan_create_guru(prec, kind, (M0, si, so), [(N1b, N2, N2), (N2, 1, 1)], 1, 0, 1, 0)
        pin, pout = (x, y) if kind < 0 else (y, x)
        line('far stage (axis 0) %s' % tag, timeit(lambda: eng.execute_ptr(h, pin.data_ptr(), pout.data_ptr(), 1.0)), alg1)
        eng.plan_destroy(h)
    del a, b, c, mid, x, y
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
    if what in ('slab', 'all'):
        run_slab('C3 (2,1,1)', 8, 256, 512, 512, 2)
        run_slab('C4@2', 8, 512, 1024, 1024, 2)
        run_slab('C4 (8,1,1)', 8, 128, 1024, 1024, 8)
        run_slab('c64 (8,1,1)', 4, 128, 1024, 1024, 8)
        if what == 'slab':
            return
    if what in ('c4', 'all'):
        run_case('C4@8', 8, False, (256, 512, 1024), 1024, (256, 1024, 512), (1024, 256, 512), 2, 4, None, 4, 4)
    if what in ('c5', 'all'):
        run_case('C5@8e', 4, True, (512, 1024, 2048), 1025, (512, 2048, 512), (2048, 512, 512), 2, 4, [513, 512], 4, 1)
    if what in ('c5odd', 'all'):
        run_case('C5@8o', 4, True, (512, 1024, 2048), 1025, (512, 2048, 513), (2048, 512, 513), 2, 4, [513, 512], 4, 1)


if __name__ == '__main__':
    main()
