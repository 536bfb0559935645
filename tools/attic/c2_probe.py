#!/usr/bin/env python3
"""Developer tool: per-pass times of single-axis plans (four-step 2^20, strided fp32 passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from mpi4py_fft_amd import fftw, _lib
from mpi4py_fft_amd.array import DeviceArray


def case(name, shape, dt, axes):
    a = DeviceArray(shape, dt)
    t = a.tensor
    (torch.view_as_real(t) if t.is_complex() else t).view(-1)[: 1 << 28].normal_()
    real = np.dtype(dt).kind == 'f'
    p = (fftw.rfftn if real else fftw.fftn)(a, axes=axes)
    b = p.output_array
    for _ in range(3):
        p.execute_scaled(a, b, 1.0)
    _lib.set_option('profile', 1)
    for _ in range(10):
        p.execute_scaled(a, b, 1.0)
    torch.cuda.synchronize()
    _lib.set_option('profile', 0)
    print(name, p.describe() if hasattr(p, 'describe') else '')
    tot = 0
    for fam, nbytes, ms, n in p.profile():
        if n:
            tot += ms / n
            print('   %-24s %8.3f ms  %7.1f GB/s' % (fam, ms / n, nbytes / (ms / n) / 1e6))
    print('   total %.3f ms' % tot, flush=True)
    p.destroy()
    del a, b, p
    torch.cuda.empty_cache()


print(torch.cuda.get_device_name(0))
which = sys.argv[1:] or ['c2', 'f32']
if 'c2' in which:
    case('C2 2^20 c128 B=64', (64, 1 << 20), 'D', (1,))
    case('2^20 c64 B=128', (128, 1 << 20), 'F', (1,))
    case('2^16 c128 B=1024', (1024, 1 << 16), 'D', (1,))
    case('2^22 c128 B=16', (16, 1 << 22), 'D', (1,))
if 'f32v' in which:
    for v in (0, 1, 2):
        _lib.set_option('variant_cols', v)
        print('--- cols variant', v)
        case('(1024,1024,1024) axis1 c64', (1024, 1024, 1024), 'F', (1,))
        case('(1024,1024,1024) axis0 c64', (1024, 1024, 1024), 'F', (0,))
        case('(1024,1024,513) axis1 c64', (1024, 1024, 513), 'F', (1,))
        case('(2048,512,1024) axis1 c64', (2048, 512, 1024), 'F', (1,))
        case('(2048,512,513) axis1 c64', (2048, 512, 513), 'F', (1,))
        case('(4096,256,1024) axis1 c64', (4096, 256, 1024), 'F', (1,))
    _lib.set_option('variant_cols', 0)
if 'f32' in which:
    case('(512,2048,513) axis1 c64', (512, 2048, 513), 'F', (1,))
    case('(2048,512,513) axis0 c64', (2048, 512, 513), 'F', (0,))
    case('(1024,1024,1024) axis1 c64', (1024, 1024, 1024), 'F', (1,))
    case('(1024,1024,1024) axis0 c64', (1024, 1024, 1024), 'F', (0,))
    case('(1024,1024,1024) axis2 c64', (1024, 1024, 1024), 'F', (2,))
    case('(1024,1024,1024) axis0 c128', (1024, 1024, 1024), 'D', (0,))
