#!/usr/bin/env python3
"""Developer probe: the serial stages of the pipelined C4@8 transform (chunk-major exchange buffers,
guru plans) timed on one GPU against the staged path's fused-split plans of the same shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mpi4py_fft_amd import _lib
from mpi4py_fft_amd.pipeline import Layout, _Stage
from mpi4py_fft_amd.libfft import FFT


def timeit(fn, iters=8, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(iters):
        s.record(); fn(); e.record(); e.synchronize()
        ts.append(s.elapsed_time(e))
    return min(ts)


def stage(name, shape, axis, lin, lout, K):
    st = _Stage(shape, axis, lin, lout, -1, 8)
    assert st.plan is not None, name
    n = 1
    for d in shape:
        n *= d
    a = torch.randn(n, dtype=torch.complex128, device='cuda')
    b = torch.empty_like(a)
    eng = _lib.engine()

    def run():
        for c in range(st.nchunks):
            eng.execute_ptr(st.plan, a.data_ptr() + c * st.step_in * 16, b.data_ptr() + c * st.step_out * 16, 1.0)
    t = timeit(run)
    f = FFT(shape, axes=(axis,), dtype='D')
    t0 = timeit(lambda: f.forward())
    print('%-44s pipelined %d chunk(s) %7.3f ms  %7.1f GB/s | natural plan %7.3f ms' % (
        name, st.nchunks, t, 2 * n * 16 / t / 1e6, t0), flush=True)
    f.destroy()
    st.destroy()


print(torch.cuda.get_device_name(0))
for K in (1, 2, 4, 8):
    s0, s1, s2 = (256, 512, 1024), (256, 1024, 512), (1024, 256, 512)
    stage('C4@8 stage 0 rows  natural -> T0(p=2,f=0)', s0, 2, Layout(s0), Layout(s0, 2, 2, 0, K), K)
    stage('C4@8 stage 1 axis1 T0(p=2,f=0) -> T1(p=4,f=2)', s1, 1, Layout(s1, 1, 2, 0, K), Layout(s1, 1, 4, 2, K), K)
    stage('C4@8 stage 2 axis0 T1(p=4,f=2) -> natural', s2, 0, Layout(s2, 0, 4, 2, K), Layout(s2), K)
# slab grid (8,1,1): stages axis 2 and axis 1 local, then T(1->0, p=8, f=2)
for K in (1, 4):
    s1, s2 = (128, 1024, 1024), (1024, 128, 1024)
    stage('slab stage 1 axis1 natural -> T(p=8,f=2)', s1, 1, Layout(s1), Layout(s1, 1, 8, 2, K), K)
    stage('slab stage 2 axis0 T(p=8,f=2) -> natural', s2, 0, Layout(s2, 0, 8, 2, K), Layout(s2), K)
