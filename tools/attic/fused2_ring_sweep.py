import sys, os
sys.path.insert(0, os.getcwd())
import torch
from mpi4py_fft_amd import PFFT, comm, _lib
_lib.set_option('fuse2', 1)
combos = [(8, 4), (12, 6), (12, 5), (12, 7), (14, 7), (14, 6), (14, 8), (13, 6), (11, 6)]
from mpi4py_fft_amd import newDistArray
# (a PFFT owns its planned arrays: three at a time, all on the SAME caller arrays, the first combination in every batch)
base = PFFT(comm.COMM_SELF, (1024,) * 3, dtype='D')
u, w = newDistArray(base, False), newDistArray(base, True)
base.destroy(); del base
import gc; gc.collect(); torch.cuda.empty_cache()
torch.view_as_real(u.tensor).normal_()
tot = {}
import gc
for b in range(0, len(combos) - 1, 2):
    batch = [combos[0]] + combos[1 + b: 3 + b]
    ffts = {}
    for ring, lag in batch:
        _lib.set_option('fuse2_ring', ring); _lib.set_option('fuse2_lag', lag)
        ffts[(ring, lag)] = PFFT(comm.COMM_SELF, (1024,) * 3, dtype='D')
    for rnd in range(3):
        for c in batch:
            f = ffts[c]
            f.forward(u, w); f.backward(w, u)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(8):
                f.forward(u, w); f.backward(w, u)
            e.record(); e.synchronize()
            tot.setdefault(c, []).append(s.elapsed_time(e) / 8)
    for f in ffts.values():
        f.destroy()
    del ffts, f
    gc.collect()
    torch.cuda.empty_cache()
for c in combos:
    print('nt ring %2d lag %2d: %s  mean %.3f ms per step' % (c + (' '.join('%.3f' % t for t in tot[c]), sum(tot[c]) / len(tot[c]))), flush=True)
