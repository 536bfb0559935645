#!/usr/bin/env python3
"""Developer survey: forward time and algorithmic HBM rate of PFFT / serial plans over the
BASELINE configs' single-GPU pieces (runs on the GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from mpi4py_fft_amd import PFFT, comm, fftw, _lib
from mpi4py_fft_amd.array import DeviceArray


def timeit(fn, iters=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(iters):
        s.record(); fn(); e.record(); e.synchronize()
        ts.append(s.elapsed_time(e))
    return min(ts)


def fill(a):
    t = a.tensor
    r = torch.view_as_real(t) if t.is_complex() else t
    r.view(-1)[: min(r.numel(), 1 << 28)].normal_()


def pfft_case(name, shape, dt, **kw):
    fft = PFFT(comm.COMM_SELF, shape, dtype=dt, **kw)
    fill(fft.forward.input_array)
    tf = timeit(lambda: fft.forward())
    tb = timeit(lambda: fft.backward())
    fl, by = fft.cost()
    print('%-46s fwd %8.3f ms %7.1f GB/s %7.1f GFLOP/s | bwd %8.3f ms %7.1f GB/s' % (
        name, tf, by / tf / 1e6, fl / tf / 1e6, tb, by / tb / 1e6), flush=True)
    fft.destroy()
    del fft
    torch.cuda.empty_cache()


def plan_case(name, shape, dt, axes):
    a = DeviceArray(shape, dt)
    fill(a)
    real = np.dtype(dt).kind == 'f'
    p = (fftw.rfftn if real else fftw.fftn)(a, axes=axes)
    b = p.output_array
    t = timeit(lambda: p.execute_scaled(a, b, 1.0))
    fl, by, nl = p.cost()
    print('%-46s     %8.3f ms %7.1f GB/s %7.1f GFLOP/s  (%d launches)' % (name, t, by / t / 1e6, fl / t / 1e6, nl), flush=True)
    p.destroy()
    del a, b, p
    torch.cuda.empty_cache()


print(torch.cuda.get_device_name(0))
pfft_case('C1  PFFT 64^3 c128', (64,) * 3, 'D')
pfft_case('    PFFT 256^3 c128', (256,) * 3, 'D')
pfft_case('C3' + "' PFFT 512^3 c128 (1 GPU)", (512,) * 3, 'D')
pfft_case('C4  PFFT 1024^3 c128', (1024,) * 3, 'D')
pfft_case('    PFFT 1024^3 c64', (1024,) * 3, 'F')
pfft_case('    PFFT 1024^3 r2c f64', (1024,) * 3, 'd')
pfft_case('C5' + "' PFFT 1024^3 r2c f32", (1024,) * 3, 'f')
pfft_case('    PFFT 2048x1024x1024 r2c f32', (2048, 1024, 1024), 'f')
pfft_case('    PFFT 1024^3 c128 padded 1.5 (683->1024)', (683, 683, 683), 'D', padding=[1.5, 1.5, 1.5])
pfft_case('    PFFT 768^3 c128 (R=12 kernels)', (768,) * 3, 'D')
pfft_case('    PFFT 384^3 c128', (384,) * 3, 'D')
pfft_case('    PFFT 768^3 r2c f64', (768,) * 3, 'd')
pfft_case('    PFFT 1536x768x768 c128', (1536, 768, 768), 'D')
pfft_case('    PFFT 512^3 r2c f64 padded 1.5 (-> 768^3)', (512,) * 3, 'd', padding=[1.5, 1.5, 1.5])
pfft_case('    PFFT 512^3 c128 padded 1.5 (-> 768^3)', (512,) * 3, 'D', padding=[1.5, 1.5, 1.5])
pfft_case('    PFFT 1024x512x512 r2c f32 padded 1.5', (1024, 512, 512), 'f', padding=[1.5, 1.5, 1.5])
pfft_case('    PFFT 1152^3 c64', (1152,) * 3, 'F')
pfft_case('    PFFT 1000^3 c64 (R=20 kernels)', (1000,) * 3, 'F')
pfft_case('    PFFT 1000^3 c128', (1000,) * 3, 'D')
pfft_case('    PFFT 640^3 r2c f64', (640,) * 3, 'd')
pfft_case('    PFFT 960^3 c128 (15 x 16 x 4, one pass)', (960,) * 3, 'D')
pfft_case('    PFFT 896^3 c128 (7 x 16 x 8, one pass) ', (896,) * 3, 'D')
plan_case('C2  batched 1-D 2^20 c128, B=64', (64, 1 << 20), 'D', (1,))
plan_case('    batched 1-D 2^20 c64, B=128', (128, 1 << 20), 'F', (1,))
plan_case('    1-D 2^24-ish: 4096x4096 rows c128', (4096, 4096), 'D', (1,))
plan_case('    4096x4096 cols c128', (4096, 4096), 'D', (0,))
plan_case('C4@8 local: (256,512,1024) axis2 c128', (256, 512, 1024), 'D', (2,))
plan_case('C4@8 local: (256,1024,512) axis1 c128', (256, 1024, 512), 'D', (1,))
plan_case('C4@8 local: (1024,256,512) axis0 c128', (1024, 256, 512), 'D', (0,))
plan_case('C5@8 local: (512,1024,2048) r2c f32 axis2', (512, 1024, 2048), 'f', (2,))
plan_case('C5@8 local: (512,2048,513) axis1 c64', (512, 2048, 513), 'F', (1,))
plan_case('C5@8 local: (2048,512,513) axis0 c64', (2048, 512, 513), 'F', (0,))
