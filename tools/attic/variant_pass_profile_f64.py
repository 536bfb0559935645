#!/usr/bin/env python3
"""Developer probe: per-pass times of the 1024^3 c128 PFFT under the fp64 strided-kernel variants
(0 default R16/T16; 3 = R16/T8 two workgroups per CU; 13 / 14 = non-temporal stores / loads+stores)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mpi4py_fft_amd import PFFT, comm, _lib

print(torch.cuda.get_device_name(0))
for v in (0, 3, 13, 14, 0, 3):
    _lib.set_option('variant_cols', v)
    fft = PFFT(comm.COMM_SELF, (1024,) * 3, dtype='D')
    torch.view_as_real(fft.forward.input_array.tensor).normal_()
    for _ in range(2):
        fft.forward(); fft.backward()
    _lib.set_option('profile', 1)
    for _ in range(5):
        fft.forward(); fft.backward()
    torch.cuda.synchronize()
    _lib.set_option('profile', 0)
    out = []
    for name, p in zip(('fwd', 'bwd'), fft._fused_plans):
        for fam, nbytes, ms, n in p.profile():
            if n:
                out.append('%s %s %.3f' % (name, fam.split()[0][5:], ms / n))
    print('variant %2d: %s' % (v, ' | '.join(out)), flush=True)
    fft.destroy()
_lib.set_option('variant_cols', 0)
