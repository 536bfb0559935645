#!/usr/bin/env python3
"""Developer probe: variants of the packed-real row kernels (r2c / c2r along the contiguous axis; option variant_rows, read at
planning), plans alternating on the SAME arrays, 5 rounds x 10 executions.
usage: real_rows_variant_probe.py <dtype d|f> <v1,v2,...> <n0xn1xn2> ..."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mpi4py_fft_amd import _lib

eng = _lib.engine()
dt = sys.argv[1]
values = [int(v) for v in sys.argv[2].split(',')]
prec, rdt, cdt = (8, torch.float64, torch.complex128) if dt == 'd' else (4, torch.float32, torch.complex64)
print(torch.cuda.get_device_name(0), dt)
for case in sys.argv[3:]:
    shape = [int(x) for x in case.split('x')]
    oshape = shape[:-1] + [shape[-1] // 2 + 1]
    a = torch.empty(shape, dtype=rdt, device='cuda').normal_()
    b = torch.empty(oshape, dtype=cdt, device='cuda')
    c = torch.empty(shape, dtype=rdt, device='cuda')
    for kind, name, src, dst in ((-2, 'r2c', a, b), (+2, 'c2r', b, c)):
        plans = {}
        for v in values:
            _lib.set_option('variant_rows', v)
            plans[v] = eng.plan_create(shape if kind == -2 else oshape, oshape if kind == -2 else shape, [len(shape) - 1], kind, prec)
        _lib.set_option('variant_rows', 0)
        tot = {v: [] for v in values}
        for rnd in range(5):
            for v in values:
                eng.execute_ptr(plans[v], src.data_ptr(), dst.data_ptr(), 1.0)
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(10):
                    eng.execute_ptr(plans[v], src.data_ptr(), dst.data_ptr(), 1.0)
                e.record(); e.synchronize()
                tot[v].append(s.elapsed_time(e) / 10)
        _lib.check_async()
        nbytes = a.numel() * a.element_size() + b.numel() * b.element_size()
        print('%-18s %s  ' % (case, name) + '  '.join('v%-2d %.3f ms %.2f' % (v, sum(tot[v]) / 5, nbytes / (sum(tot[v]) / 5) / 8e9) for v in values), flush=True)
        for h in plans.values():
            eng.plan_destroy(h)
    del a, b, c
