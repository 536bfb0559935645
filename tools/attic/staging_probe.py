#!/usr/bin/env python3
"""Developer probe: host <-> device rates of DeviceArray assignment / np.asarray (pinned bounce buffers),
of a caller-owned pinned array (host_empty), and of the plain pageable torch copy they replace."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from mpi4py_fft_amd import empty, host_empty


def best(fn, n=3):
    ts = []
    for _ in range(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return min(ts)


print(torch.cuda.get_device_name(0), 'host threads', torch.get_num_threads())
for gib in (1, 4):
    n = int(gib * 2 ** 30) // 16
    shape = (n,)
    h = np.ones(shape, dtype='D')
    u = empty(shape, 'D')
    gb = n * 16 / 1e9
    print('%d GiB  pageable torch copy   H2D %6.1f GB/s  D2H %6.1f GB/s' % (
        gib, gb / best(lambda: u.tensor.copy_(torch.from_numpy(h))), gb / best(lambda: u.tensor.cpu())), flush=True)
    print('%d GiB  staged (u[...] = h)   H2D %6.1f GB/s  D2H %6.1f GB/s' % (
        gib, gb / best(lambda: u.__setitem__(Ellipsis, h)), gb / best(lambda: np.asarray(u))), flush=True)
    p = host_empty(shape, 'D')
    p[...] = h
    print('%d GiB  pinned host array     H2D %6.1f GB/s  D2H %6.1f GB/s' % (
        gib, gb / best(lambda: u.__setitem__(Ellipsis, p)), gb / best(lambda: u.get(out=p))), flush=True)
    del h, u, p
