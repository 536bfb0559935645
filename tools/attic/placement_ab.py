#!/usr/bin/env python3
"""Developer probe (round 4): how much of the headline's box-to-box / run-to-run spread belongs to WHICH physical memory the
arrays got?  tools/probes/placement_probe shows buffers that write at 5.8 TB/s next to buffers that write at 6.9 TB/s in one
process.  Here: the 1024^3 complex128 step (a) on arrays as they come, (b) on the fastest-writing three of ten candidates
while the slowest ones stay allocated (so that the library's workspace cannot land on them either)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mpi4py_fft_amd import PFFT, comm, _lib, newDistArray
from mpi4py_fft_amd.array import DeviceArray

n = 1024
nbytes = n ** 3 * 16

def step_time(fft, u, w, reps=3):
    ts = []
    for _ in range(reps):
        fft.forward(u, w); fft.backward(w, u)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10):
            fft.forward(u, w); fft.backward(w, u)
        e.record(); e.synchronize()
        ts.append(s.elapsed_time(e) / 10)
    return ts

def write_rate(t):
    best = 0
    for _ in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); t.fill_(1); e.record(); e.synchronize()
        best = max(best, t.numel() / s.elapsed_time(e) / 1e6)
    return best

print(torch.cuda.get_device_name(0), flush=True)
for trial in range(2):
    fft = PFFT(comm.COMM_SELF, (n,) * 3, dtype='D')
    u, w = newDistArray(fft, False), newDistArray(fft, True)
    torch.view_as_real(u.tensor).normal_()
    print('trial %d, arrays as they come: %s ms per step; write rates u %.0f w %.0f GB/s' % (trial, ' '.join('%.3f' % t for t in step_time(fft, u, w)),
          write_rate(torch.view_as_real(w.tensor)), write_rate(torch.view_as_real(u.tensor))), flush=True)
    fft.destroy(); del fft, u, w
    import gc; gc.collect(); torch.cuda.empty_cache(); _lib.lib().gfft_scratch_release()
    cands = [torch.empty(nbytes, dtype=torch.uint8, device='cuda') for _ in range(10)]
    rates = [write_rate(c) for c in cands]
    order = sorted(range(10), key=lambda i: -rates[i])
    print('   candidates (write GB/s):', ' '.join('%.0f' % r for r in rates), flush=True)
    keep = [cands[i] for i in order[:2]]
    hoard = [cands[i] for i in order[5:]]           # the slow half stays allocated: nothing else can land there
    del cands
    gc.collect(); torch.cuda.empty_cache()
    fft = PFFT(comm.COMM_SELF, (n,) * 3, dtype='D')
    u = DeviceArray((n,) * 3, dtype='D', tensor=keep[0].view(torch.complex128).reshape(n, n, n))
    w = DeviceArray((n,) * 3, dtype='D', tensor=keep[1].view(torch.complex128).reshape(n, n, n))
    torch.view_as_real(u.tensor).normal_()
    print('trial %d, fastest-writing candidates, slow half hoarded: %s ms per step' % (trial, ' '.join('%.3f' % t for t in step_time(fft, u, w))), flush=True)
    fft.destroy(); del fft, u, w, keep, hoard
    gc.collect(); torch.cuda.empty_cache(); _lib.lib().gfft_scratch_release()
