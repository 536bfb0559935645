#!/usr/bin/env python3
"""Developer probe: clean A/B of COMBINATIONS of libgfft planning options inside a cubic complex128 PFFT -- one plan
set per combination, all run alternately on the SAME caller arrays (placement moves the step time by several per
cent), 5 rounds x 10 steps; then the per-pass times of each.
usage: ab_combo_probe.py [-n 1024 | -n 1024x1024x2048] [-d D] "fuse2=1" "fuse2=0,wtile=0" ...   (the first one is the baseline)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mpi4py_fft_amd import PFFT, comm, _lib
if os.environ.get('GFFT_AB_LIB'):          # (a build next to libgfft.so: A/B of compile-time choices, process against process)
    _lib.LIBPATH = os.path.join(os.path.dirname(_lib.LIBPATH), os.environ['GFFT_AB_LIB'])

args = sys.argv[1:]
n, dt = 1024, 'D'
while args and args[0].startswith('-'):
    if args[0] == '-n': n = tuple(int(x) for x in args[1].split('x')) if 'x' in args[1] else int(args[1])
    if args[0] == '-d': dt = args[1]
    args = args[2:]
combos = [dict((kv.split('=')[0], int(kv.split('=')[1])) for kv in a.split(',') if kv) for a in args]
keys = sorted({k for c in combos for k in c})
base = {k: combos[0].get(k, 0) for k in keys}
print(torch.cuda.get_device_name(0), n, dt, flush=True)
ffts = []
for c in combos:
    for k in keys:
        _lib.set_option(k, c.get(k, base[k]))
    ffts.append(PFFT(comm.COMM_SELF, n if isinstance(n, tuple) else (n,) * 3, dtype=dt))
for k in keys:
    _lib.set_option(k, base[k])
u, w = ffts[0].forward.input_array, ffts[0].forward.output_array
(torch.view_as_real(u.tensor) if u.tensor.is_complex() else u.tensor).normal_()
tot = [[] for _ in combos]
# every combination computes the same transform: forward outputs against the baseline's
ref = None
for i, f in enumerate(ffts):
    f.forward(u, w)
    torch.cuda.synchronize()
    if ref is None:
        ref = w.tensor.clone()
    else:
        d = float((torch.view_as_real(w.tensor) - torch.view_as_real(ref)).abs().max().item())
        m = float(torch.view_as_real(ref).abs().max().item())
        print('%-40s forward max|diff| vs baseline %.2e (max|ref| %.2e)' % (args[i], d, m), flush=True)
    f.backward(w, u)
del ref
for rnd in range(5):
    for i, f in enumerate(ffts):
        f.forward(u, w); f.backward(w, u)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10):
            f.forward(u, w); f.backward(w, u)
        e.record(); e.synchronize()
        tot[i].append(s.elapsed_time(e) / 10)
for i, c in enumerate(combos):
    m = sum(tot[i]) / len(tot[i])
    print('%-40s %s  mean %.3f ms per step (%+.2f %%)' % (args[i], ' '.join('%.3f' % t for t in tot[i]), m,
                                                        100 * (m / (sum(tot[0]) / len(tot[0])) - 1)), flush=True)
_lib.set_option('profile', 1)
for i, f in enumerate(ffts):
    for _ in range(5):
        f.forward(u, w)
    torch.cuda.synchronize()
    if f._fused_plans:
        print('%-40s forward passes:' % args[i], ', '.join('%s %.3f ms' % (name, ms / max(k, 1)) for name, nb, ms, k in f._fused_plans[0].profile()), flush=True)
    for _ in range(5):
        f.backward(w, u)
    torch.cuda.synchronize()
    if f._fused_plans:
        print('%-40s backward passes:' % args[i], ', '.join('%s %.3f ms' % (name, ms / max(k, 1)) for name, nb, ms, k in f._fused_plans[1].profile()), flush=True)
_lib.set_option('profile', 0)
