#!/usr/bin/env python3
"""Developer probe: variants of the stand-alone strided pass (option variant_cols, read at planning) on single-axis plans,
plans alternating on the SAME arrays, 5 rounds x 10 executions.   usage: cols_variant_probe.py <dtype D|F> <v1,v2,...> <case> ...
case = n0xn1xn2:axis"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mpi4py_fft_amd import _lib

eng = _lib.engine()
dt = sys.argv[1]
values = [int(v) for v in sys.argv[2].split(',')]
prec, cdt, isz = (8, torch.complex128, 16) if dt == 'D' else (4, torch.complex64, 8)
print(torch.cuda.get_device_name(0), dt)
OPT = os.environ.get("PROBE_OPT", "variant_cols")
for case in sys.argv[3:]:
    shp, ax = case.split(':')
    shape = [int(x) for x in shp.split('x')]
    a = torch.empty(shape, dtype=cdt, device='cuda')
    torch.view_as_real(a).normal_()
    b = torch.empty_like(a)
    plans = {}
    for v in values:
        _lib.set_option(OPT, v)
        plans[v] = eng.plan_create(shape, shape, [int(ax)], -1, prec)
    _lib.set_option(OPT, 0)
    tot = {v: [] for v in values}
    for rnd in range(5):
        for v in values:
            eng.execute_ptr(plans[v], a.data_ptr(), b.data_ptr(), 1.0)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(10):
                eng.execute_ptr(plans[v], a.data_ptr(), b.data_ptr(), 1.0)
            e.record(); e.synchronize()
            tot[v].append(s.elapsed_time(e) / 10)
    _lib.check_async()
    nbytes = 2 * a.numel() * isz
    print('%-22s' % case + '  '.join('v%-2d %.3f ms %.2f' % (v, sum(tot[v]) / 5, nbytes / (sum(tot[v]) / 5) / 8e9) for v in values), flush=True)
    for h in plans.values():
        eng.plan_destroy(h)
    del a, b
