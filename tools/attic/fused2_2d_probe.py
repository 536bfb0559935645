import sys, os
sys.path.insert(0, os.getcwd())
import torch
from mpi4py_fft_amd import fftw, zeros, _lib
def timeit(fn, iters=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(iters):
        s.record(); fn(); e.record(); e.synchronize(); ts.append(s.elapsed_time(e))
    return min(ts)
a = zeros((512, 1024, 1024), 'D'); torch.view_as_real(a.tensor).normal_()
out = zeros((512, 1024, 1024), 'D')
for fuse in (0, 1, 0, 1):
    _lib.set_option('fuse2', fuse)
    p = fftw.fftn(a, axes=(1, 2), output_array=out)
    q = fftw.ifftn(out, axes=(1, 2), output_array=a)
    print('2-D 512 x 1024^2 c128 fuse2=%d: fwd %.3f ms  bwd %.3f ms' % (fuse, timeit(lambda: p.execute_scaled(a, out, 1.0)), timeit(lambda: q.execute_scaled(out, a, 1.0 / 1048576))), flush=True)
    p.destroy(); q.destroy()
