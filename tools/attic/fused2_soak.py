#!/usr/bin/env python3
"""Developer aid: soak the fused pass pairs -- 1024^3 c128 forward / backward and C2, N repetitions each on
fresh random data now and then, every result compared bit for bit with the first one of its data set and the
round trip with the input (a hand-off that raced once in a thousand launches would show here)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mpi4py_fft_amd import PFFT, comm, fftw, zeros, _lib
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
fft = PFFT(comm.COMM_SELF, (1024,) * 3, dtype='D')
u, w = fft.forward.input_array, fft.forward.output_array
bad = 0
for rnd in range(4):
    torch.view_as_real(u.tensor).normal_()
    keep = u.tensor.clone()
    first = fft.forward(u, w).tensor.clone()
    for k in range(reps // 4):
        fft.forward(u, w)
        bad += int(not torch.equal(w.tensor, first))
        fft.backward(w, u)
        err = (u.tensor - keep).abs().max().item()
        bad += int(err > 1e-12 * 8)
        u.tensor.copy_(keep)
    print('1024^3 round %d: mismatches so far %d' % (rnd, bad), flush=True)
fft.destroy()
a = zeros((64, 1 << 20), 'D'); torch.view_as_real(a.tensor).normal_()
p = fftw.fftn(a, axes=(1,))
first = p.execute_scaled(a, p.output_array, 1.0).tensor.clone()
for k in range(reps * 5):
    p.execute_scaled(a, p.output_array, 1.0)
    bad += int(not torch.equal(p.output_array.tensor, first))
print('C2: mismatches in total %d' % bad)
print('soak OK' if bad == 0 else 'soak FAILED')
