#!/usr/bin/env python3
"""Runs named serial plans / PFFTs a few times each, for tools/prof.sh (rocprofv3 passes).
usage: prof_cases.py case [case...]   cases: r2c_d r2c_f c5rows c5ax1 c5ax0 pad_d pad_f c2 cols4096 cols2048"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gc
import numpy as np
import torch
from mpi4py_fft_amd import PFFT, comm, fftw
from mpi4py_fft_amd.array import DeviceArray
from mpi4py_fft_amd.libfft import FFT

REPS = 6


def fill(t):
    (torch.view_as_real(t) if t.is_complex() else t).view(-1)[:1 << 28].normal_()


def serial(shape, dt, axes, padding=False):
    f = FFT(shape, axes=axes, dtype=dt, padding=padding)
    fill(f.forward.input_array.tensor)
    for _ in range(REPS):
        f.forward()
        f.backward()
    torch.cuda.synchronize()
    f.destroy()


def pfft(shape, dt, **kw):
    f = PFFT(comm.COMM_SELF, shape, dtype=dt, **kw)
    fill(f.forward.input_array.tensor)
    for _ in range(REPS):
        f.forward()
        f.backward()
    torch.cuda.synchronize()
    f.destroy()


CASES = {
    'r2c_d': lambda: pfft((1024,) * 3, 'd'),
    'r2c_f': lambda: pfft((1024,) * 3, 'f'),
    'r2c_d2048': lambda: pfft((1024, 1024, 2048), 'd'),
    'c960': lambda: pfft((960, 960, 960), 'D'),
    'c960f': lambda: pfft((960, 960, 960), 'F'),
    'r960': lambda: pfft((960, 960, 960), 'd'),
    'c5rows': lambda: serial((512, 1024, 2048), 'f', (2,)),
    'c5ax1': lambda: serial((512, 2048, 513), 'F', (1,)),
    'c5ax0': lambda: serial((2048, 512, 513), 'F', (0,)),
    'pad_d': lambda: pfft((683,) * 3, 'D', padding=[1.5] * 3),
    'pad_f': lambda: pfft((1024, 512, 512), 'f', padding=[1.5] * 3),
    'c2': lambda: serial((64, 1 << 20), 'D', (1,)),
    'c2f': lambda: serial((128, 1 << 20), 'F', (1,)),
    'cols4096': lambda: serial((4096, 4096), 'D', (0,)),
    'cols2048': lambda: serial((2048, 2048, 4), 'D', (0,)),
    'mix3f': lambda: serial((512, 1536, 512), 'F', (1,)),
    'mix5f': lambda: serial((512, 1280, 512), 'F', (1,)),
    'mix3f8': lambda: serial((256, 3072, 512), 'F', (1,)),
    'cols4096f': lambda: serial((4096, 4096, 8), 'F', (0,)),
}
for name in sys.argv[1:]:
    CASES[name]()
    gc.collect()
    torch.cuda.empty_cache()
print('done', sys.argv[1:])
