#!/usr/bin/env python3
"""Developer probe: does the step time of the 1024^3 c128 PFFT drift with how long the GPU has been busy?
(per-block averages of consecutive forward+backward steps, HIP events)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mpi4py_fft_amd import PFFT, comm

print(torch.cuda.get_device_name(0))
fft = PFFT(comm.COMM_SELF, (1024,) * 3, dtype='D')
torch.view_as_real(fft.forward.input_array.tensor).normal_()
torch.cuda.synchronize()
t_start = time.time()
for block in range(16):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10):
        fft.forward(); fft.backward()
    e.record(); e.synchronize()
    print('t = %5.2f s  steps %3d-%3d: %.3f ms per step' % (time.time() - t_start, block * 10, block * 10 + 9, s.elapsed_time(e) / 10), flush=True)
print('idle 3 s'); time.sleep(3)
for block in range(3):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10):
        fft.forward(); fft.backward()
    e.record(); e.synchronize()
    print('after idle: %.3f ms per step' % (s.elapsed_time(e) / 10), flush=True)
