import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from mpi4py_fft_amd import fftw, zeros, _lib, PFFT, comm
def timeit(fn, iters=8, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(iters):
        s.record(); fn(); e.record(); e.synchronize(); ts.append(s.elapsed_time(e))
    return min(ts), sorted(ts)[len(ts)//2]
for fuse in (0, 1):
    _lib.set_option('fuse2', fuse)
    fft = PFFT(comm.COMM_SELF, (1024,)*3, dtype='D')
    torch.view_as_real(fft.forward.input_array.tensor).normal_()
    print(_lib.engine().plan_describe(fft._fused_plans[0]._plan))
    print('fuse2=%d 1024^3 c128 fwd %.3f / %.3f ms   bwd %.3f / %.3f ms' % ((fuse,) + timeit(lambda: fft.forward()) + timeit(lambda: fft.backward())), flush=True)
    fft.destroy(); del fft; torch.cuda.empty_cache()
    a = zeros((64, 1 << 20), 'D'); torch.view_as_real(a.tensor).normal_()
    p = fftw.fftn(a, axes=(1,))
    print('fuse2=%d C2 64 x 2^20 c128 fwd %.3f / %.3f ms' % ((fuse,) + timeit(lambda: p.execute_scaled(a, p.output_array, 1.0))), flush=True)
    p.destroy(); del a, p; torch.cuda.empty_cache()
if len(sys.argv) > 1:
    _lib.set_option('fuse2', int(sys.argv[1]))
    for ring, lag in ((6, 3), (8, 4), (8, 6), (12, 6), (16, 4), (16, 8), (16, 12)):
        _lib.set_option('fuse2_ring', ring); _lib.set_option('fuse2_lag', lag)
        fft = PFFT(comm.COMM_SELF, (1024,)*3, dtype='D')
        print('ring %2d lag %d: fwd %.3f / %.3f  bwd %.3f / %.3f' % ((ring, lag) + timeit(lambda: fft.forward()) + timeit(lambda: fft.backward())), flush=True)
        fft.destroy(); del fft; torch.cuda.empty_cache()
if len(sys.argv) > 2 and sys.argv[2] == 'c2':
    for ring, lag in ((3, 1), (4, 2), (6, 2), (8, 2), (6, 3), (8, 4), (12, 4)):
        _lib.set_option('fuse2_ring', ring); _lib.set_option('fuse2_lag', lag)
        a = zeros((64, 1 << 20), 'D'); torch.view_as_real(a.tensor).normal_()
        p = fftw.fftn(a, axes=(1,))
        print('C2 ring %2d lag %d: %.3f / %.3f ms' % ((ring, lag) + timeit(lambda: p.execute_scaled(a, p.output_array, 1.0), iters=20)), flush=True)
        p.destroy(); del a, p; torch.cuda.empty_cache()
