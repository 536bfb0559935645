#!/usr/bin/env python3
"""Developer probe: cols-kernel variants on padded / natural layouts, 1024^3 c128."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mpi4py_fft_amd import _lib
from tools.layout_probe import run_pass

n = 1024
nbytes = n * n * n * 16
P = n + 16
a = torch.empty(n * n * P * 2, dtype=torch.float64, device='cuda').normal_()
b = torch.empty_like(a)


def near(pin, pout):
    return [n, n, 1, n, n * pin, 0, 1, pin, n * pout, 0, 1, pout]


def far(pin, pout):
    return [n, 1, n, n, 0, pin, 1, n * pin, 0, pout, 1, n * pout]


print(torch.cuda.get_device_name(0))
vs = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else '0,2,3,9,11,12,5,10').split(',')]
for v in vs:
    r = []
    for nm, g, dst in (('near pad inplace', near(P, P), a), ('far pad inplace', far(P, P), a),
                       ('near pad->nat', near(P, n), b), ('near nat->pad', near(n, P), b),
                       ('far nat->nat', far(n, n), b)):
        t = run_pass(g, v, a, dst)
        r.append('%s %.3f' % (nm, t))
    print('variant %2d: %s' % (v, ' | '.join(r)), flush=True)
