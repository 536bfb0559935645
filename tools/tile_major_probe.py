#!/usr/bin/env python3
"""Developer probe (round 5): what a TILE-MAJOR workspace would be worth to the stand-alone strided fp64 pass of the 1024^3
schedule.  The pass of the plan reads the caller's natural array (256-byte segments, 16 KiB apart) and writes pitched workspace
rows (256-byte segments, 16.6 KiB apart); here the same kernel writes [plane][tile of 16 columns][row][16] instead -- every
tile one contiguous 256 KiB run (gfft_plan_set_tiles on the output side) -- and, the mirror case, READS such a buffer and writes
natural rows.  Stand-alone guru plans, same buffers, alternating."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mpi4py_fft_amd import _lib

n = 1024
P = n + 16
eng = _lib.engine()
src = torch.randn(n * n * n, dtype=torch.complex128, device='cuda')
dst = torch.empty(n * n * P + (1 << 20), dtype=torch.complex128, device='cuda')


def timeit(h, a, b, reps=8):
    for _ in range(2):
        eng.execute_ptr(h, a.data_ptr(), b.data_ptr(), 1.0)
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); eng.execute_ptr(h, a.data_ptr(), b.data_ptr(), 1.0); e.record(); e.synchronize()
        ts.append(s.elapsed_time(e))
    return min(ts), sum(ts) / len(ts)


plans = {}
# axis 1 of [i0][i1][c]: natural in (stride n between i1), out rows pitched P apart  (what plan_fused3's first pass does)
plans['natural -> pitched rows'] = (eng.plan_create_guru(8, -1, (n, n, P), [(n, n * n, n * P), (n, 1, 1)]), src, dst)
# ... out tile-major: [i0][tile][i1][16]
h = eng.plan_create_guru(8, -1, (n, n, 16), [(n, n * n, n * P), (n, 1, 1)])
assert eng.plan_set_tiles(h, 1, 16, n * 16)
plans['natural -> tile-major'] = (h, src, dst)
# mirror: read pitched rows / tile-major, write natural
plans['pitched rows -> natural'] = (eng.plan_create_guru(8, -1, (n, P, n), [(n, n * P, n * n), (n, 1, 1)]), dst, src)
h = eng.plan_create_guru(8, -1, (n, 16, n), [(n, n * P, n * n), (n, 1, 1)])
assert eng.plan_set_tiles(h, 0, 16, n * 16)
plans['tile-major -> natural'] = (h, dst, src)
print(torch.cuda.get_device_name(0))
for rnd in range(3):
    for name, (h, a, b) in plans.items():
        lo, av = timeit(h, a, b)
        print('round %d  %-26s best %.3f ms  mean %.3f ms  = %.0f GB/s of 2 S' % (rnd, name, lo, av, 2 * n ** 3 * 16 / lo / 1e6), flush=True)
