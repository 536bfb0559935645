#!/usr/bin/env python3
"""Developer probe (round 5): what bench.py's admission gates cost at 1024^3 on one GPU, step by step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mpi4py_fft_amd import PFFT, comm, selftest

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
fft = PFFT(comm.COMM_SELF, (n, n, n), dtype='D')
u = fft.forward.input_array.tensor
torch.view_as_real(u).normal_()
u0 = u.clone()


def T(label, fn):
    torch.cuda.synchronize(); t = time.perf_counter(); r = fn(); torch.cuda.synchronize()
    print('%-40s %.3f s' % (label, time.perf_counter() - t), flush=True)
    return r


for rep in range(2):
    T('exchange_check', lambda: selftest.exchange_check(fft, comm.COMM_SELF))
    T('  one transfer_check', lambda: selftest.transfer_check(fft.transfer[0], 1, u.device))
    T('empty_cache', torch.cuda.empty_cache)
    out = T('forward', lambda: fft.forward().tensor)
    print(T('forward_gate', lambda: selftest.forward_gate(fft, comm.COMM_SELF, u0, out)))
    T('fingerprint', lambda: selftest.fingerprint(out))
    T('backward', lambda: fft.backward())
