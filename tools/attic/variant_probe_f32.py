#!/usr/bin/env python3
"""Developer probe: fp32 cols-kernel variants (GFFT_VARIANT_COLS) on 1024^3 / (512,2048,1024) c64,
pitched and natural layouts -- re-run of the two-workgroups-per-CU A/B after the 4-byte LDS rule."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mpi4py_fft_amd import _lib

L = _lib.lib()


def run_pass(geom, variant, a, b, iters=8):
    g = (ctypes.c_int64 * 12)(*geom)
    st = _lib.current_stream()
    fn = lambda: _lib.check(L.gfft_debug_pass(g, 4, 1, variant, 0, a.data_ptr(), b.data_ptr(), st))
    fn(); fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(iters):
        s.record(); fn(); e.record(); e.synchronize()
        ts.append(s.elapsed_time(e))
    return min(ts)


print(torch.cuda.get_device_name(0))
for n, n0 in ((1024, 1024), (512, 2048), (2048, 512)):
    w = 1024
    P = w + 32
    a = torch.empty(n0 * n * P * 2 + 64, dtype=torch.float32, device='cuda').normal_()
    b = torch.empty_like(a)
    near = lambda pi, po: [n, n0, 1, w, n * pi, 0, 1, pi, n * po, 0, 1, po]
    far = lambda pi, po: [n, 1, n0, w, 0, pi, 1, n0 * pi, 0, po, 1, n0 * po]
    gb = n0 * n * w * 16 / 1e6
    for v in ([0, 2, 3] if n == 1024 else [0]):
        r = []
        for nm, g, dst in (('near pad inplace', near(P, P), a), ('far pad inplace', far(P, P), a),
                           ('near nat->nat', near(w, w), b), ('far nat->nat', far(w, w), b)):
            t = run_pass(g, v, a, dst)
            r.append('%s %.3f ms %4.0f GB/s' % (nm, t, gb / t))
        print('n=%d (x%d x%d) variant %d: %s' % (n, n0, w, v, ' | '.join(r)), flush=True)
    del a, b
