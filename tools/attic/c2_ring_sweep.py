#!/usr/bin/env python3
"""Developer probe: config C2 (64 x 2^20 complex128, batched 1-D) under ring / lag settings of its fused four-step
launch, plans alternating on the same arrays."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mpi4py_fft_amd import fftw, zeros, _lib
combos = [(0, 0), (12, 6), (16, 8), (10, 5), (14, 7), (12, 4), (8, 4), (20, 10), (12, 8), (16, 6)]
a = zeros((64, 1 << 20), 'D'); torch.view_as_real(a.tensor).normal_()
plans = {}
for ring, lag in combos:
    _lib.set_option('fuse2_ring', ring); _lib.set_option('fuse2_lag', lag)
    plans[(ring, lag)] = fftw.fftn(a, axes=(1,))
_lib.set_option('fuse2_ring', 0); _lib.set_option('fuse2_lag', 0)
tot = {c: [] for c in combos}
for rnd in range(5):
    for c in combos:
        p = plans[c]
        p.execute_scaled(a, p.output_array, 1.0)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20):
            p.execute_scaled(a, p.output_array, 1.0)
        e.record(); e.synchronize()
        tot[c].append(s.elapsed_time(e) / 20)
print(torch.cuda.get_device_name(0))
for c in combos:
    print('C2 ring %2d lag %2d: %s  mean %.4f ms' % (c + (' '.join('%.4f' % t for t in tot[c]), sum(tot[c]) / 5)), flush=True)
