#!/usr/bin/env python3
"""Developer probe: fixed caller arrays, the library's 16 GiB workspace re-allocated at different
physical places (a dummy block of varying size is held while it is allocated); 20 steps each.
Then the reverse: fixed workspace, fresh caller arrays."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mpi4py_fft_amd import PFFT, comm, _lib
from mpi4py_fft_amd.array import DeviceArray

n = 1024
print(torch.cuda.get_device_name(0))
fft = PFFT(comm.COMM_SELF, (n,) * 3, dtype='D')
torch.view_as_real(fft.forward.input_array.tensor).normal_()


def steps(u=None, v=None, k=20):
    a = (u, v) if u is not None else ()
    b = (v, u) if u is not None else ()
    for _ in range(2):
        fft.forward(*a); fft.backward(*b)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(k):
        fft.forward(*a); fft.backward(*b)
    e.record(); e.synchronize()
    return s.elapsed_time(e) / k


L = _lib.lib()
print('first workspace: %.3f ms per step' % steps(), flush=True)
for i, gib in enumerate((3, 0, 7, 1, 12, 5, 0, 20)):
    torch.cuda.synchronize()
    L.gfft_scratch_release()
    dummy = torch.empty(max(1, gib << 30), dtype=torch.uint8, device='cuda')
    t = steps(k=2)                      # allocates the workspace while the dummy is held
    del dummy
    torch.cuda.empty_cache()
    print('workspace re-allocated behind a %2d GiB block: %.3f ms per step' % (gib, steps()), flush=True)
print('fixed workspace, fresh caller arrays:')
for i in range(6):
    hold = torch.empty((i * 3 + 1) << 30, dtype=torch.uint8, device='cuda')
    u = DeviceArray((n,) * 3, 'D'); v = DeviceArray((n,) * 3, 'D')
    del hold
    torch.view_as_real(u.tensor).normal_()
    print('   arrays at %#x / %#x: %.3f ms per step' % (u.tensor.data_ptr(), v.tensor.data_ptr(), steps(u, v)), flush=True)
    del u, v
    torch.cuda.empty_cache()
