#!/usr/bin/env python3
"""Developer probe (round 4): the fused pair's kernel VARIANTS (option fuse2 = 1: 16 values per thread, radices
16.16.4, two LDS exchanges, 1024 threads; 3: 32 values per thread, radices 32.32, one exchange, 512 threads) --
correctness of each against numpy on a small plane count, then a clean A/B on the same caller arrays at 1024^3
and on C2 (64 x 2^20).   usage: fused2_variant_ab.py [v1,v2,...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from mpi4py_fft_amd import PFFT, comm, _lib, fftw, zeros

values = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else '1,3').split(',')]
print(torch.cuda.get_device_name(0), 'fuse2 variants', values, flush=True)

# -- correctness: (1024, 40, 1024) c128 all axes, and 32 x 2^20, against numpy
rng = np.random.default_rng(11)
shape = (1024, 40, 1024)
x = rng.standard_normal(shape) + 1j * rng.standard_normal(shape)
ref = np.fft.fftn(x)
y = rng.standard_normal((32, 1 << 20)) + 1j * rng.standard_normal((32, 1 << 20))
refy = np.fft.fft(y, axis=1)
for v in values:
    _lib.set_option('fuse2', v)
    a = zeros(shape, 'D'); f = fftw.fftn(a, axes=(0, 1, 2)); b = fftw.ifftn(f.output_array, axes=(0, 1, 2), output_array=zeros(shape, 'D'))
    desc = _lib.engine().plan_describe(f._plan)
    a[...] = x
    got = np.asarray(f.execute_scaled(a, f.output_array, 1.0))
    back = np.asarray(b.execute_scaled(f.output_array, b.output_array, 1.0 / x.size))
    print('fuse2=%d 3-D: fused=%s fwd err %.2e  round trip %.2e' % (v, 'fused pair' in desc, np.abs(got - ref).max() / np.abs(ref).max(),
                                                                     np.abs(back - x).max() / np.abs(x).max()), flush=True)
    f.destroy(); b.destroy(); del a, f, b
    a = zeros(y.shape, 'D'); f = fftw.fftn(a, axes=(1,))
    desc = _lib.engine().plan_describe(f._plan)
    a[...] = y
    got = np.asarray(f.execute_scaled(a, f.output_array, 1.0))
    print('fuse2=%d 2^20: fused=%s fwd err %.2e' % (v, 'fused pair' in desc, np.abs(got - refy).max() / np.abs(refy).max()), flush=True)
    f.destroy(); del a, f
    torch.cuda.empty_cache()
del x, ref, y, refy

def ev_time(fn, n):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record(); e.synchronize()
    return s.elapsed_time(e) / n

# -- A/B at 1024^3 c128 on the same caller arrays
n = 1024
ffts = {}
for v in values:
    _lib.set_option('fuse2', v)
    ffts[v] = PFFT(comm.COMM_SELF, (n,) * 3, dtype='D')
u, w = ffts[values[0]].forward.input_array, ffts[values[0]].forward.output_array
torch.view_as_real(u.tensor).normal_()
tot = {v: [] for v in values}
for rnd in range(5):
    for v in values:
        f = ffts[v]
        f.forward(u, w); f.backward(w, u)
        tot[v].append(ev_time(lambda: (f.forward(u, w), f.backward(w, u)), 10))
for v in values:
    print('1024^3 c128 fuse2=%d: %s  mean %.3f ms per fwd+bwd step' % (v, ' '.join('%.3f' % t for t in tot[v]), sum(tot[v]) / len(tot[v])), flush=True)
for v in values:
    _lib.set_option('profile', 1)
    f = ffts[v]
    for _ in range(5):
        f.forward(u, w)
    torch.cuda.synchronize()
    try:
        print('fuse2=%d forward passes (ms):' % v, ['%.3f' % t for t in f._fused_plans[0].profile()], flush=True)
    except Exception as ex:
        print('profile n/a:', ex)
    _lib.set_option('profile', 0)
for v in values:
    ffts[v].destroy()
del ffts, u, w
torch.cuda.empty_cache()

# -- C2
a = zeros((64, 1 << 20), 'D'); torch.view_as_real(a.tensor).normal_()
plans = {}
for v in values:
    _lib.set_option('fuse2', v)
    plans[v] = fftw.fftn(a, axes=(1,))
tot = {v: [] for v in values}
for rnd in range(5):
    for v in values:
        p = plans[v]
        p.execute_scaled(a, p.output_array, 1.0)
        tot[v].append(ev_time(lambda: p.execute_scaled(a, p.output_array, 1.0), 20))
for v in values:
    print('C2 64 x 2^20 c128 fuse2=%d: %s  mean %.4f ms' % (v, ' '.join('%.4f' % t for t in tot[v]), sum(tot[v]) / len(tot[v])), flush=True)
_lib.set_option('fuse2', 1)
