#!/usr/bin/env python3
"""Developer probe (round 5): is the penalty of the FAR strided pass of the 3-D schedules a matter of how far its rows lie apart?
The in-place strided pass on a pitched workspace of n x n rows, three ways, same buffer: lines over the NEAR index (rows P entries
apart), over the FAR index (n P apart), and over the far index of a BLOCKED layout W[k/B][i][k%B][c] (rows B P apart, B = 16)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mpi4py_fft_amd import _lib

eng = _lib.engine()
B = 16


def timeit(h, a, reps=8):
    for _ in range(2):
        eng.execute_ptr(h, a.data_ptr(), a.data_ptr(), 1.0)
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); eng.execute_ptr(h, a.data_ptr(), a.data_ptr(), 1.0); e.record(); e.synchronize()
        ts.append(s.elapsed_time(e))
    return min(ts), sum(ts) / len(ts)


print(torch.cuda.get_device_name(0))
for n, prec, P in ((1024, 8, 1040), (768, 8, 776), (1024, 4, 1056), (512, 8, 528)):
    dt = torch.complex128 if prec == 8 else torch.complex64
    w = torch.randn(n * n * P, dtype=dt, device='cuda')
    plans = {
        'near  (rows P apart)': eng.plan_create_guru(prec, -1, (n, P, P), [(n, n * P, n * P), (n, 1, 1)]),
        'far   (rows n P apart)': eng.plan_create_guru(prec, -1, (n, n * P, n * P), [(n, P, P), (n, 1, 1)]),
        'blocked (rows 16 P apart)': eng.plan_create_guru(prec, -1, (n, B * P, B * P), [(n // B, n * B * P, n * B * P), (B, P, P), (n, 1, 1)]),
    }
    for rnd in range(2):
        for name, h in plans.items():
            lo, av = timeit(h, w)
            print('n = %d %s  round %d  %-26s best %.3f ms  mean %.3f ms  = %.0f GB/s of 2 S' %
                  (n, 'c128' if prec == 8 else 'c64', rnd, name, lo, av, 2 * n ** 3 * 2 * prec / lo / 1e6), flush=True)
    del w
    torch.cuda.empty_cache()
