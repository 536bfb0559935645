#!/usr/bin/env python3
"""Developer probe: clean A/B of one libgfft planning option inside a cubic PFFT -- one plan set per value,
all run alternately on the SAME caller arrays (their placement moves the step time by several per cent,
DESIGN section 6), 5 rounds x 10 steps.   usage: ab_option_probe.py <option> <v1,v2,...> [dtype] [n]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mpi4py_fft_amd import PFFT, comm, _lib

opt = sys.argv[1]
values = [int(x) for x in sys.argv[2].split(',')]
dt = sys.argv[3] if len(sys.argv) > 3 else 'D'
n = int(sys.argv[4]) if len(sys.argv) > 4 else 1024
default = {'grid_cap': 0, 'xcd_swizzle': -1}.get(opt, 0)
print(torch.cuda.get_device_name(0), opt, dt, n)
ffts = {}
for v in values:
    _lib.set_option(opt, v)
    ffts[v] = PFFT(comm.COMM_SELF, (n,) * 3, dtype=dt)
_lib.set_option(opt, default)
u, w = ffts[values[0]].forward.input_array, ffts[values[0]].forward.output_array
(torch.view_as_real(u.tensor) if u.tensor.is_complex() else u.tensor).normal_()
tot = {v: [] for v in values}
for rnd in range(5):
    for v in values:
        f = ffts[v]
        if opt == 'grid_cap':
            _lib.set_option(opt, v)          # read at launch time
        f.forward(u, w); f.backward(w, u)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10):
            f.forward(u, w); f.backward(w, u)
        e.record(); e.synchronize()
        tot[v].append(s.elapsed_time(e) / 10)
_lib.set_option(opt, default)
for v in values:
    print('%s = %5d: %s  mean %.3f ms per step' % (opt, v, ' '.join('%.3f' % t for t in tot[v]), sum(tot[v]) / len(tot[v])), flush=True)
