#!/usr/bin/env python3
"""Developer probe (round 5): clean A/B of option combinations on ONE serial plan (fftw.fftn over given axes): one plan per
combination, all on the same arrays, alternating; results checked against the first combination's.
usage: serial_ab_probe.py 128x1048576 F 1  "fuse2_f32=0" "fuse2_f32=1" ...     (shape, dtype, axes comma separated)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from mpi4py_fft_amd import fftw, _lib, zeros

shape = tuple(int(x) for x in sys.argv[1].split('x'))
dt = sys.argv[2]
axes = tuple(int(x) for x in sys.argv[3].split(','))
args = sys.argv[4:]
combos = [dict((kv.split('=')[0], int(kv.split('=')[1])) for kv in a.split(',') if kv) for a in args]
keys = sorted({k for c in combos for k in c})
base = {k: combos[0].get(k, 0) for k in keys}
a = zeros(shape, dt)
out = zeros(shape, dt)
torch.view_as_real(a.tensor).normal_()
plans = []
for c in combos:
    for k in keys:
        _lib.set_option(k, c.get(k, base[k]))
    plans.append(fftw.fftn(a, axes=axes, output_array=out))
for k in keys:
    _lib.set_option(k, base[k])
print(torch.cuda.get_device_name(0), shape, dt, axes, flush=True)
ref = None
for i, p in enumerate(plans):
    p.execute_scaled(a, out, 1.0)
    torch.cuda.synchronize()
    if ref is None:
        ref = out.tensor.clone()
    else:
        d = float((torch.view_as_real(out.tensor) - torch.view_as_real(ref)).abs().max().item())
        print('%-44s max|diff| vs first %.2e (max|ref| %.2e)' % (args[i], d, float(torch.view_as_real(ref).abs().max().item())))
tot = [[] for _ in combos]
for rnd in range(5):
    for i, p in enumerate(plans):
        p.execute_scaled(a, out, 1.0)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20):
            p.execute_scaled(a, out, 1.0)
        e.record(); e.synchronize()
        tot[i].append(s.elapsed_time(e) / 20)
nbytes = 2 * a.tensor.numel() * a.tensor.element_size()
for i, c in enumerate(combos):
    m = sum(tot[i]) / len(tot[i])
    print('%-44s %s  mean %.4f ms (%+.2f %%)  %.0f GB/s of 2 S' % (args[i], ' '.join('%.4f' % t for t in tot[i]), m,
          100 * (m / (sum(tot[0]) / len(tot[0])) - 1), nbytes / m / 1e6), flush=True)
    print('    ', _lib.engine().plan_describe(plans[i]._plan).strip().split('\n')[-1].strip()[:200])
