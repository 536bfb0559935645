#!/usr/bin/env python3
"""Developer probe (round 5): per-pass times of a one-rank 3-D plan against the offset of libgfft's workspace relative to the
caller's (2 MiB-aligned) arrays (option ws_skew_kib).   usage: skew_sweep.py 1024x1024x2048 d  k1,k2,...   (KiB)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mpi4py_fft_amd import PFFT, comm, _lib

shape = tuple(int(x) for x in sys.argv[1].split('x'))
dt = sys.argv[2]
skews = [int(x) for x in sys.argv[3].split(',')]
f = PFFT(comm.COMM_SELF, shape, dtype=dt)
t = f.forward.input_array.tensor
(torch.view_as_real(t) if t.is_complex() else t).normal_()
print('in %#x out %#x' % (f.forward.input_array.data_ptr, f.forward.output_array.data_ptr))
for skew in skews:
    _lib.set_option('ws_skew_kib', skew)
    f.forward(); f.backward()
    _lib.set_option('profile', 1)
    for _ in range(4):
        f.forward()
    torch.cuda.synchronize()
    a = f._fused_plans[0].profile()
    for _ in range(4):
        f.backward()
    torch.cuda.synchronize()
    b = f._fused_plans[1].profile()
    _lib.set_option('profile', 0)
    fw = [ms / max(k, 1) for name, nb, ms, k in a]
    bw = [ms / max(k, 1) for name, nb, ms, k in b]
    print('skew %6d KiB  fwd %s = %.3f   bwd %s = %.3f' % (skew, ' '.join('%.3f' % x for x in fw), sum(fw), ' '.join('%.3f' % x for x in bw), sum(bw)), flush=True)
_lib.set_option('ws_skew_kib', 0)
