#!/usr/bin/env python3
"""Developer tool: per-pass kernel times of a PFFT forward/backward (HIP events inside libgfft)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mpi4py_fft_amd import PFFT, comm, _lib

def run(shape, dt, **kw):
    fft = PFFT(comm.COMM_SELF, shape, dtype=dt, **kw)
    t = fft.forward.input_array.tensor
    (torch.view_as_real(t) if t.is_complex() else t).view(-1)[: 1 << 28].normal_()
    for _ in range(2):
        fft.forward(); fft.backward()
    _lib.set_option('profile', 1)
    for _ in range(5):
        fft.forward(); fft.backward()
    torch.cuda.synchronize()
    _lib.set_option('profile', 0)
    plans = list(fft._fused_plans) if fft._fused_plans else [x.fwd for x in fft.xfftn] + [x.bck for x in fft.xfftn]
    print('%s %s %s' % (shape, dt, kw))
    for name, p in zip(('fwd', 'bwd') if fft._fused_plans else [str(i) for i in range(len(plans))], plans):
        for fam, nbytes, ms, n in p.profile():
            if n:
                print('   %s %-20s %8.3f ms  %7.1f GB/s' % (name, fam, ms / n, nbytes / (ms / n) / 1e6))
    fft.destroy()

print(torch.cuda.get_device_name(0))
if len(sys.argv) > 1 and sys.argv[1] == 'padded':
    for fuse in (True, False):
        run((1024, 512, 512), 'f', padding=[1.5] * 3, fuse=fuse)
        run((512, 512, 512), 'd', padding=[1.5] * 3, fuse=fuse)
    sys.exit(0)
run((1024,) * 3, 'D')
run((1024,) * 3, 'd')
run((1024,) * 3, 'F')
run((1024,) * 3, 'f')
run((512,) * 3, 'D')
run((2048, 1024, 1024), 'f')
