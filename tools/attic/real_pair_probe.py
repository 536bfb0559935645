#!/usr/bin/env python3
"""Developer probe (round 4): the fused pairs of REAL transforms (fuse2_kinds bits 32 / 64) against the unfused plans,
on the same caller arrays: 1024^3 r2c / c2r fp64 and (1024, 1024, 2048)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mpi4py_fft_amd import PFFT, comm, _lib

print(torch.cuda.get_device_name(0), flush=True)
for shape in ((1024, 1024, 1024), (1024, 1024, 2048)):
    ffts = {}
    for kinds in (30, 126):
        _lib.set_option('fuse2_kinds', kinds)
        ffts[kinds] = PFFT(comm.COMM_SELF, shape, dtype='d')
    _lib.set_option('fuse2_kinds', 126)
    u, w = ffts[30].forward.input_array, ffts[30].forward.output_array
    u.tensor.normal_()
    tf = {k: [] for k in ffts}
    tb = {k: [] for k in ffts}
    for rnd in range(5):
        for k, f in ffts.items():
            f.forward(u, w); f.backward(w, u)
            for which, fn in ((tf, lambda: f.forward(u, w)), (tb, lambda: f.backward(w, u))):
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(10):
                    fn()
                e.record(); e.synchronize()
                which[k].append(s.elapsed_time(e) / 10)
    for k in ffts:
        print('%s r2c f64, pairs %s: forward %.3f ms  backward %.3f ms' % (shape, 'fused' if k == 126 else 'unfused', sum(tf[k]) / 5, sum(tb[k]) / 5), flush=True)
    _lib.set_option('profile', 1)
    for k, f in ffts.items():
        for _ in range(5):
            f.forward(u, w)
        torch.cuda.synchronize()
        print('   forward passes:', ', '.join('%s %.3f ms' % (n, ms / max(c, 1)) for n, nb, ms, c in f._fused_plans[0].profile()), flush=True)
        for _ in range(5):
            f.backward(w, u)
        torch.cuda.synchronize()
        print('   backward passes:', ', '.join('%s %.3f ms' % (n, ms / max(c, 1)) for n, nb, ms, c in f._fused_plans[1].profile()), flush=True)
    _lib.set_option('profile', 0)
    for f in ffts.values():
        f.destroy()
    del ffts, u, w
    torch.cuda.empty_cache()
