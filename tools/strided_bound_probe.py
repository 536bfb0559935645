#!/usr/bin/env python3
"""Developer probe (round 5): what bounds the stand-alone strided fp64 pass of the 1024^3 schedule.  Inside the 3-D plan, same
caller arrays, plans alternating: the default kernel (32 values per thread, one exchange, 512 threads), the rounds 1-3 kernel
(variant 17), and -- from a library whose fft_pow2_f64 was built with -DGFFT_VARIANTS -- the ACCESS PATTERN ALONE (variant 10:
the same 16-column x 1024-row tiles loaded and stored, no butterflies, no LDS exchange), next to the streaming copy of the
same bytes on the same box (gfft_probe_copy).   usage: GFFT_AB_LIB=libgfft_var.so python tools/strided_bound_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpi4py_fft_amd import _lib
if os.environ.get('GFFT_AB_LIB'):
    _lib.LIBPATH = os.path.join(os.path.dirname(_lib.LIBPATH), os.environ['GFFT_AB_LIB'])
import torch
from mpi4py_fft_amd import PFFT, comm

n = 1024
variants = [0, 17, 10, 14]
ffts = {}
for v in variants:
    _lib.set_option('variant_cols', v)
    try:
        ffts[v] = PFFT(comm.COMM_SELF, (n,) * 3, dtype='D')
    except Exception as e:
        print('variant', v, 'not available:', repr(e)[:100])
_lib.set_option('variant_cols', 0)
u, w = ffts[0].forward.input_array, ffts[0].forward.output_array
torch.view_as_real(u.tensor).normal_()
L, st = _lib.lib(), _lib.current_stream()
nbytes = u.tensor.numel() * 16
res = {v: [] for v in ffts}
copies = []
for rnd in range(4):
    for v, f in ffts.items():
        f.forward(u, w)
        _lib.set_option('profile', 1)
        for _ in range(5):
            f.forward(u, w)
        torch.cuda.synchronize()
        _lib.set_option('profile', 0)
        res[v].append({name: ms / max(k, 1) for name, nb, ms, k in f._fused_plans[0].profile()})
    for _ in range(2):
        _lib.check(L.gfft_probe_copy(u.tensor.data_ptr(), w.tensor.data_ptr(), nbytes, st))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        _lib.check(L.gfft_probe_copy(u.tensor.data_ptr(), w.tensor.data_ptr(), nbytes, st))
    e1.record(); torch.cuda.synchronize()
    copies.append(e0.elapsed_time(e1) / 5)
print(torch.cuda.get_device_name(0))
print('streaming copy of 2 x 16 GiB (gfft_probe_copy): %s ms  = %.0f GB/s' % (' '.join('%.3f' % c for c in copies), 2 * nbytes / min(copies) / 1e6))
for v in ffts:
    names = list(res[v][0])
    for nm in names:
        ts = [r[nm] for r in res[v]]
        print('variant_cols %2d  %-32s %s ms   best = %.0f GB/s of 2 S' % (v, nm, ' '.join('%.3f' % t for t in ts), 2 * nbytes / min(ts) / 1e6))
