#!/usr/bin/env python3
"""Developer aid: time of the fused launch of the 1024^3 complex128 forward transform, alone, under
GFFT_FUSE2_DEBUG = 3 (tickets and counters only), 4 (A tiles only), 5 (B tiles only) -- results are
wrong by construction in those modes; run one mode per process."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mpi4py_fft_amd import PFFT, comm, _lib
fft = PFFT(comm.COMM_SELF, (1024,) * 3, dtype='D')
torch.view_as_real(fft.forward.input_array.tensor).normal_()
err = os.dup(2); null = os.open(os.devnull, os.O_WRONLY); os.dup2(null, 2)
for _ in range(2):
    fft.forward()
_lib.set_option('profile', 1)
for _ in range(6):
    fft.forward()
torch.cuda.synchronize()
_lib.set_option('profile', 0)
os.dup2(err, 2)
for fam, nbytes, ms, n in fft._fused_plans[0].profile():
    if n:
        print('debug=%s  %-40s %8.3f ms' % (os.environ.get('GFFT_FUSE2_DEBUG', '0'), fam, ms / n), flush=True)
