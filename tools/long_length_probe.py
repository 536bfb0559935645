#!/usr/bin/env python3
"""Developer probe (round 5): single 1-D transforms of 2^24 entries and beyond (plan.cpp plan_long: one strided pass in
front of a four-step transform of the cofactor) -- time per transform, the passes of the plan, GB/s of one read + one write."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mpi4py_fft_amd import fftw, _lib

print(torch.cuda.get_device_name(0))
for n, dt in ((1 << 22, 'D'), (1 << 24, 'D'), (1 << 25, 'D'), (3 << 23, 'D'), (1 << 27, 'D'), (1 << 30, 'D'), (1 << 27, 'F'), (1 << 30, 'F'),
              (10 ** 8, 'D'), (16777259, 'D')):
    a = fftw.aligned((n,), dtype=dt)
    out = fftw.aligned((n,), dtype=dt)
    torch.view_as_real(a.tensor).normal_()
    plan = fftw.fftn(a, axes=(0,), output_array=out)
    for _ in range(2):
        plan()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); plan(); e.record(); e.synchronize()
        ts.append(s.elapsed_time(e))
    esz = 16 if dt == 'D' else 8
    best = min(ts)
    print('n = %d (%s): %.3f ms  = %.0f GB/s of one read + one write' % (n, 'complex128' if dt == 'D' else 'complex64', best, 2 * n * esz / best / 1e6), flush=True)
    print(_lib.engine().plan_describe(plan._plan), flush=True)
    plan.destroy()
    del a, out
    torch.cuda.empty_cache()
