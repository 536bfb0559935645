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
for shape in ((256, 512, 512), (512, 512, 512)):
    a = zeros(shape, 'D'); torch.view_as_real(a.tensor).normal_()
    out = zeros(shape, 'D')
    for rnd in range(2):
        for v in (1, 2):
            _lib.set_option('fuse2_n512', v)
            p = fftw.fftn(a, axes=(1, 2), output_array=out)
            fused = 'fused pair' in _lib.engine().plan_describe(p._plan)
            print('2-D %s c128 fuse2_n512=%d fused=%s: %.3f ms' % (shape, v, fused, timeit(lambda: p.execute_scaled(a, out, 1.0))), flush=True)
            p.destroy()
_lib.set_option('fuse2_n512', 1)
