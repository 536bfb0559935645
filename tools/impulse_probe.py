#!/usr/bin/env python3
"""Developer probe (round 6): impulse responses of the serial plans (tests/cases.py impulse_errors) in units of eps *
log2 n, per kernel family, precision and mapping -- what the thresholds of tests/test_gpu_rounding_guard.py were set
from -- and the same with one twiddle entry off by 1e-9 (fp64) / 1e-5 (fp32) through the test hook debug_tw_exp."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests import cases
from mpi4py_fft_amd import _lib

LENGTHS = [16, 32, 64, 128, 256, 512, 1024, 2048, 4096, 48, 96, 192, 384, 768, 1536, 3072, 20, 40, 80, 160, 640, 1000, 4000,
           240, 480, 960, 112, 896, 1792, 22, 26, 34, 121, 343, 1331, 8192, 16384, 9216, 521, 1009, 4099]
for dt in 'DF':
    eps = cases.EPS[dt]
    for strided in (False, True):
        worst = 0
        for n in LENGTHS:
            try:
                ef, eb = cases.impulse_errors(n, dt, strided)
            except Exception as e:
                print(dt, 'strided' if strided else 'rows', n, 'FAILED', str(e)[:100])
                continue
            u = eps * np.log2(n)
            print('%s %-7s n=%-6d fwd %.2e (%.2f eps log2 n)  bwd %.2e (%.2f)' % (dt, 'strided' if strided else 'rows', n, ef, ef / u, eb, eb / u), flush=True)
# one wrong entry
for dt, e in (('D', 9), ('F', 5)):
    for n in (16, 64, 512, 1024, 2048, 768, 640, 960):
        for idx in (1, n // 2 + 3, n // 3):
            _lib.set_option('debug_tw_index', idx)
            _lib.set_option('debug_tw_exp', e)
            try:
                ef, eb = cases.impulse_errors(n, dt, False)
                es, _ = cases.impulse_errors(n, dt, True)
            finally:
                _lib.set_option('debug_tw_exp', 0)
            u = cases.EPS[dt] * np.log2(n)
            print('%s n=%-5d entry %-4d off by 1e-%d: rows fwd %.2e (%.1f eps log2 n) bwd %.2e, strided fwd %.2e' % (dt, n, idx, e, ef, ef / u, eb, es), flush=True)
