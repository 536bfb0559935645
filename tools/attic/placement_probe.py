#!/usr/bin/env python3
"""Developer probe: does the step time of the 1024^3 c128 PFFT depend on WHERE its arrays lie relative
to each other?  One PFFT; the caller's input / output arrays carved from one block at chosen offsets,
the library's workspace started `ws_skew_kib` into its buffer.  20 steps per setting."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mpi4py_fft_amd import PFFT, comm, _lib
from mpi4py_fft_amd.array import DeviceArray

n = 1024
N = n ** 3
print(torch.cuda.get_device_name(0))
MAXSKEW = 8 << 20
big = torch.empty(2 * N * 16 + 4 * MAXSKEW, dtype=torch.uint8, device='cuda')
fft = PFFT(comm.COMM_SELF, (n,) * 3, dtype='D')


def carve(off):
    t = big[off:off + N * 16].view(torch.complex128).view(n, n, n)
    return DeviceArray((n,) * 3, 'D', tensor=t)


def run(in_off, out_off, skew_kib, steps=20):
    _lib.set_option('ws_skew_kib', skew_kib)
    u, v = carve(in_off), carve(N * 16 + MAXSKEW + out_off)
    for _ in range(2):
        fft.forward(u, v); fft.backward(v, u)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(steps):
        fft.forward(u, v); fft.backward(v, u)
    e.record(); e.synchronize()
    return s.elapsed_time(e) / steps


torch.view_as_real(carve(0).tensor).normal_()
run(0, 0, MAXSKEW >> 10, 2)            # the workspace buffer at its largest first: later skews reuse it
for skew in (0, 4, 16, 64, 128, 256, 512, 1024, 2048, 4096, 1, 0):
    print('in +0, out +0, workspace +%5d KiB: %.3f ms per step' % (skew, run(0, 0, skew)), flush=True)
for off in (0, 4 << 10, 64 << 10, 256 << 10, 1 << 20, (1 << 20) + (64 << 10), 0):
    print('in +0, out +%8d B, workspace +0: %.3f ms per step' % (off, run(0, off, 0)), flush=True)
for off in (4 << 10, 256 << 10, 1 << 20):
    print('in +%8d B, out +0, workspace +0: %.3f ms per step' % (off, run(off, 0, 0)), flush=True)
_lib.set_option('ws_skew_kib', 0)
