#!/usr/bin/env python3
"""Developer probe (round 5): the far-axis flat-tile pass of a forward r2c transform with 1025-wide half-spectrum rows
((1024, 1024, 2048) float64): per-pass times on freshly allocated caller arrays, instance after instance (placement), and
under workspace skews."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mpi4py_fft_amd import PFFT, comm, _lib

shape = tuple(int(x) for x in sys.argv[1].split('x')) if len(sys.argv) > 1 else (1024, 1024, 2048)
ninst = int(sys.argv[2]) if len(sys.argv) > 2 else 5


def passes(f, reps=5, skew=None):
    _lib.set_option('profile', 1)
    for _ in range(reps):
        f.forward()
    torch.cuda.synchronize()
    out = ', '.join('%s %.3f' % (name, ms / max(k, 1)) for name, nb, ms, k in f._fused_plans[0].profile())
    for _ in range(reps):
        f.backward()
    torch.cuda.synchronize()
    out += ' | bwd: ' + ', '.join('%s %.3f' % (name, ms / max(k, 1)) for name, nb, ms, k in f._fused_plans[1].profile())
    _lib.set_option('profile', 0)
    return out


ffts = []
for i in range(ninst):
    f = PFFT(comm.COMM_SELF, shape, dtype='d')
    f.forward.input_array.tensor.normal_()
    f.forward(); f.backward()
    print('instance %d in %#x out %#x: %s' % (i, f.forward.input_array.data_ptr, f.forward.output_array.data_ptr, passes(f)), flush=True)
    ffts.append(f)
print('again, same instances:')
for i, f in enumerate(ffts):
    print('instance %d: %s' % (i, passes(f)), flush=True)
f = ffts[0]
for skew in (0, 4, 16, 64, 256, 1024, 2048, 4096, 8192 + 4):
    _lib.set_option('ws_skew_kib', skew)
    f.forward(); f.backward()
    print('ws_skew_kib %5d: %s' % (skew, passes(f)), flush=True)
_lib.set_option('ws_skew_kib', 0)
