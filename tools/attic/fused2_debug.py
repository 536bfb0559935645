#!/usr/bin/env python3
"""Developer aid: one fused launch with GFFT_FUSE2_DEBUG=1 (bounded waits, counters printed by libgfft)."""
import os, sys
os.environ.setdefault('GFFT_FUSE2_DEBUG', '1')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from mpi4py_fft_amd import fftw, zeros, _lib
shape = (1024, 16, 1024)
rng = np.random.default_rng(5)
x = rng.standard_normal(shape) + 1j * rng.standard_normal(shape)
_lib.set_option('fuse2', 0)
a = zeros(shape, 'D'); a[...] = x
f0 = fftw.fftn(a, axes=(0, 1, 2))
want = np.asarray(f0.execute_scaled(a, f0.output_array, 1.0)).copy()
_lib.set_option('fuse2', 1)
f1 = fftw.fftn(a, axes=(0, 1, 2))
print(_lib.engine().plan_describe(f1._plan), flush=True)
for rep in range(2):
    got = np.asarray(f1.execute_scaled(a, f1.output_array, 1.0))
    torch.cuda.synchronize()
    bad = np.abs(got - want) > 1e-12 * np.abs(want).max()
    print('rep %d: mismatching entries %d of %d; by i1-plane: %s' % (rep, bad.sum(), bad.size, bad.reshape(shape).sum(axis=(0, 2))), flush=True)
b0 = fftw.ifftn(f1.output_array, axes=(0, 1, 2), output_array=zeros(shape, 'D'))
back = np.asarray(b0.execute_scaled(f1.output_array, b0.output_array, 1.0 / x.size))
print('backward (fused): max err vs input %.3e' % np.abs(back - x).max())
p = fftw.fftn(zeros((32, 1 << 20), 'D'), axes=(1,))
print(_lib.engine().plan_describe(p._plan))
p.execute_scaled(p.input_array, p.output_array, 1.0)
torch.cuda.synchronize()
print('done')
