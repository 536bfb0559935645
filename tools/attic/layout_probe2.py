#!/usr/bin/env python3
"""Developer probe 2: pitch / grid-cap / schedule sweeps for the 1024^3 c128 strided passes."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mpi4py_fft_amd import _lib
from tools.layout_probe import run_pass, L

n = 1024
N2 = n * n
nbytes = n * n * n * 16
a = torch.empty(n * n * (n + 256) * 2, dtype=torch.float64, device='cuda').normal_()
b = torch.empty_like(a)


def rep(name, t):
    print('%-60s %8.3f ms  %7.1f GB/s' % (name, t, 2 * nbytes / t / 1e6), flush=True)


def near(pin, pout):   # axis-1 pass, pitches in elements
    return [n, n, 1, n, n * pin, 0, 1, pin, n * pout, 0, 1, pout]


def far(pin, pout):    # axis-0 pass
    return [n, 1, n, n, 0, pin, 1, n * pin, 0, pout, 1, n * pout]


print(torch.cuda.get_device_name(0))
P = n + 16
for v, vn in ((0, 'FFT'), (5, 'pattern')):
    for nm, g in (('near nat->nat', near(n, n)), ('near nat->pad', near(n, P)), ('near pad->nat', near(P, n)),
                  ('near pad->pad', near(P, P)), ('far nat->nat', far(n, n)), ('far nat->pad', far(n, P)),
                  ('far pad->nat', far(P, n)), ('far pad->pad', far(P, P))):
        rep('%s %s' % (vn, nm), run_pass(g, v, a, b))
    rep('%s far pad in place' % vn, run_pass(far(P, P), v, a, a))
    rep('%s near pad in place' % vn, run_pass(near(P, P), v, a, a))
for padb in (64, 128, 256, 384, 512, 768, 1024, 1280, 2304, 4352):
    Pp = n + padb // 16
    rep('FFT far pad->pad pitch +%dB' % padb, run_pass(far(Pp, Pp), 0, a, a))
    rep('FFT near pad->nat pitch +%dB' % padb, run_pass(near(Pp, n), 0, a, b))
for cap in (256, 512, 1024, 2048, 4096, 16384, 1 << 20):
    _lib.set_option('grid_cap', cap)
    rep('FFT far pad in place cap=%d' % cap, run_pass(far(P, P), 0, a, a))
    rep('FFT near pad->nat cap=%d' % cap, run_pass(near(P, n), 0, a, b))
