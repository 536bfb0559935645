#!/usr/bin/env python3
"""Developer tool: cost of the packed-layout (split) adapters on the C4@8 / C4@4 local stages,
against the natural-layout pass plus the pack / unpack kernels they replace."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from mpi4py_fft_amd import fftw, _lib
from mpi4py_fft_amd.array import DeviceArray


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def case(shape, axis, dt, pin, pout):
    a = DeviceArray(shape, dt)
    b = DeviceArray(shape, dt)
    c = DeviceArray(shape, dt)
    torch.view_as_real(a.tensor).normal_()
    eng = _lib.engine()
    isz = np.dtype(dt).itemsize
    p = fftw.fftn(a, axes=(axis,), output_array=b)
    t_nat = timeit(lambda: p.execute_scaled(a, b, 1.0))
    t_pack = timeit(lambda: eng.pack(b.tensor, c.tensor, shape, axis, max(pin, pout, 2), isz))
    t_unpack = timeit(lambda: eng.unpack(c.tensor, b.tensor, shape, axis, max(pin, pout, 2), isz))
    res = {}
    for name, si, so in (('in', pin, 1), ('out', 1, pout), ('both', pin, pout)):
        ok = p.set_split(0, si) and p.set_split(1, so)
        res[name] = timeit(lambda: p.execute_scaled(a, b, 1.0)) if ok else float('nan')
    nbytes = 2 * a.tensor.numel() * isz
    print('%-22s axis %d %s in/%d out/%d: natural %.3f ms (%.0f GB/s) | split in %.3f out %.3f both %.3f | pack %.3f unpack %.3f'
          % (shape, axis, dt, pin, pout, t_nat, nbytes / t_nat / 1e6, res['in'], res['out'], res['both'], t_pack, t_unpack), flush=True)
    p.destroy()


print(torch.cuda.get_device_name(0))
case((256, 512, 1024), 2, 'D', 1, 2)
case((256, 1024, 512), 1, 'D', 2, 4)
case((1024, 256, 512), 0, 'D', 4, 1)
case((512, 512, 1024), 2, 'D', 1, 2)
case((512, 1024, 512), 1, 'D', 2, 2)
case((512, 1024, 1024), 1, 'D', 1, 2)
case((256, 1024, 512), 1, 'F', 2, 4)
