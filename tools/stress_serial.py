#!/usr/bin/env python3
"""Developer tool: random serial plans at the lengths the kernel tables special-case (powers of
two up to 8192, 3^b 2^k, 5^c 2^k, x7 / x11 / x13 splits, primes) in every kind and precision,
forward values against the oracle (numpy pocketfft) and round trips."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import pfft_oracle as O
from mpi4py_fft_amd import FFT, asdevice

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 120.0
rng = np.random.default_rng(seed)
LONG = [64, 128, 256, 512, 1024, 2048, 4096, 8192, 48, 96, 192, 384, 768, 1536, 3072, 1152, 20, 40, 80, 160, 320,
        640, 1000, 1280, 2000, 2560, 896, 448, 1792, 704, 1408, 832, 960, 1920, 1232, 97, 251, 1021, 2049,
        240, 480, 3840, 720, 1440, 2880, 1200, 2400, 112, 224, 3584, 7680, 1792 * 2]      # (round 5: the unequal-width stage kernels, and real lines of twice such lengths)
SHORT = [1, 2, 3, 4, 5, 6, 7, 8, 12, 16, 17, 33, 64, 65, 128, 130]
mode = sys.argv[3] if len(sys.argv) > 3 else 'small'
MID = [16, 34, 66, 100, 128, 192, 256, 257, 384, 512, 513, 640, 768, 1000, 1024, 1536, 2048, 4096, 240, 480, 960, 896, 720, 1200, 448]
t0, done = time.time(), 0
while time.time() - t0 < budget:
    nd = int(rng.integers(1, 4))
    ax_long = int(rng.integers(0, nd))
    if mode == 'mid':
        # several long axes at once: the multi-axis schedules (workspace passes, flattened tiles)
        # and wide strided passes at the sizes the big configurations have
        nd = int(rng.integers(2, 4))
        ax_long = int(rng.integers(0, nd))
        shape = tuple(int(rng.choice(MID)) for _ in range(nd))
        if not 200_000 <= np.prod(shape) <= 48_000_000:
            continue
    else:
        shape = tuple(int(rng.choice(LONG)) if i == ax_long else int(rng.choice(SHORT)) for i in range(nd))
    if np.prod(shape) > 48_000_000 or (mode != 'mid' and np.prod(shape) > 6_000_000):
        continue
    k = int(rng.integers(1, nd + 1))
    axes = tuple(int(a) for a in rng.permutation(nd)[:k])
    if rng.random() < 0.7 and ax_long not in axes:
        axes = axes[:-1] + (ax_long,) if rng.random() < 0.5 else (ax_long,) + axes[1:]
        axes = tuple(dict.fromkeys(axes))
    dt = str(rng.choice(list('dDfF')))
    if dt in 'df' and shape[axes[-1]] < 2:
        continue
    pad = False
    if len(axes) == 1 and rng.random() < 0.2:
        pad = 1.5
    fft = FFT(shape, axes, dtype=dt, padding=pad)
    ref = O.OFFT(shape, axes, dt, padding=pad)
    A = O.rng_array(shape, dt, int(rng.integers(1 << 30)))
    B = np.asarray(fft.forward(asdevice(A)))
    Bref = ref.forward(A)
    n = max(shape[a] for a in axes)
    tol = (2e-10 if dt in 'dD' else 3e-4) * max(1.0, np.log2(n) / 8)
    assert B.shape == Bref.shape and B.dtype == Bref.dtype, (shape, axes, dt, pad)
    err = np.abs(B - Bref).max() / max(np.abs(Bref).max(), 1e-30)
    assert err <= tol, (shape, axes, dt, pad, err)
    if not pad:
        A2 = np.asarray(fft.backward(asdevice(Bref)))
        rt = np.linalg.norm(A2 - A) / np.linalg.norm(A)
        assert rt <= (1e-10 if dt in 'dD' else 3e-4), (shape, axes, dt, rt)
    fft.destroy()
    done += 1
print('serial stress seed %d: %d plans checked in %.0f s' % (seed, done, time.time() - t0))
