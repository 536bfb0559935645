import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mpi4py_fft_amd import PFFT, comm, _lib
for shape, dt in (((640,) * 3, 'D'), ((640,) * 3, 'd'), ((683,) * 3, 'D')):
    for mixv in (0, 1):
        _lib.set_option('mixv', mixv)
        f = PFFT(comm.COMM_SELF, shape, dtype=dt, padding=[1.5] * 3)
        u = f.forward.input_array
        t = u.tensor
        (torch.view_as_real(t) if t.is_complex() else t).normal_()
        for _ in range(2):
            f.forward(); f.backward()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(5):
            f.forward()
        e.record(); e.synchronize(); tf = s.elapsed_time(e) / 5
        s.record()
        for _ in range(5):
            f.backward()
        e.record(); e.synchronize(); tb = s.elapsed_time(e) / 5
        print(shape, dt, 'padded to', tuple(u.shape), 'mixv', mixv, 'fwd %.3f ms bwd %.3f ms' % (tf, tb), 'one-plan' if f._fused_plans else 'staged chain', flush=True)
        f.destroy()
    _lib.set_option('mixv', 1)
