import os, sys, torch
sys.path.insert(0, '.')
from mpi4py_fft_amd import PFFT, comm, _lib
def timeit(fn, iters=8, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(iters):
        s.record(); fn(); e.record(); e.synchronize(); ts.append(s.elapsed_time(e))
    return min(ts)
for shape, dt in (((1024,)*3, 'F'), ((1024,)*3, 'f'), ((512,)*3, 'F'), ((2048, 1024, 1024), 'f')):
    r = []
    for v in (0, 2, 0, 2):
        _lib.set_option('variant_cols', v)
        f = PFFT(comm.COMM_SELF, shape, dtype=dt)
        t = f.forward.input_array.tensor
        (torch.view_as_real(t) if t.is_complex() else t).normal_()
        r.append('v%d fwd %.3f bwd %.3f' % (v, timeit(f.forward), timeit(f.backward)))
        f.destroy()
    _lib.set_option('variant_cols', 0)
    print(shape, dt, ' | '.join(r), flush=True)
