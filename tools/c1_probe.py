import sys, os
sys.path.insert(0, '/root/repo')
import torch, numpy as np
from mpi4py_fft_amd import PFFT, comm, newDistArray, _lib
for n, dt in ((64, 'D'), (128, 'D'), (64, 'F'), (32, 'D')):
    fft = PFFT(comm.COMM_SELF, (n, n, n), dtype=dt)
    p = fft._fused_plans[0]
    print(n, dt, p._eng.plan_describe(p._plan))
    u = newDistArray(fft, False)
    for _ in range(20): fft.forward(u)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(200): fft.forward(u)
    e.record(); e.synchronize()
    print('  forward %.2f us per call (200 back to back)' % (s.elapsed_time(e) / 200 * 1e3))
    # under a HIP graph
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        fft.forward(u); torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=side):
            for _ in range(20): fft.forward(u)
        g.replay(); torch.cuda.synchronize()
        s.record(side)
        for _ in range(10): g.replay()
        e.record(side); e.synchronize()
    print('  forward %.2f us per call inside a HIP graph of 20' % (s.elapsed_time(e) / 200 * 1e3))
    fft.destroy()
