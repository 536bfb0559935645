#!/usr/bin/env python3
"""Developer tool: randomised PFFT stress on one GPU (thread ranks): power-of-two and awkward
shapes, 1-8 ranks, paddings, slabs, collapse, forced relay / pack-fusion settings, vs the oracle."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests import cases

# the pipelined path on libgfft's own wire needs an RCCL that accepts several ranks per GPU: the
# in-process stand-in of tests/fake_rccl (built on demand)
import subprocess
from mpi4py_fft_amd import _lib, pipeline
_here = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'fake_rccl')
_so = os.path.join(_here, 'libfake_rccl.so')
if not os.path.exists(_so):
    subprocess.check_call(['/opt/rocm/bin/hipcc', '-O2', '-std=c++17', '-fPIC', '-shared', '-x', 'hip',
                           '--offload-arch=gfx950', os.path.join(_here, 'fake_rccl.cpp'), '-o', _so])
_lib.check_wire(_lib.lib().gfft_rccl_load(_so.encode()))
pipeline.Pipeline.MIN_CHUNK_BYTES = 0
pipeline.Pipeline.MIN_WIDTH = 4

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 120.0
rng = np.random.default_rng(seed)
mode = sys.argv[3] if len(sys.argv) > 3 else 'small'
pool = [8, 12, 13, 16, 16, 18, 24, 27, 28, 32, 32, 32, 40, 48, 56, 64, 64, 64, 96, 128, 128]
limit = 3_000_000
if mode == 'mid':
    # the lengths the big configurations run (register kernels with wide tiles, line-rounded
    # workspace passes, packed-real rows, 513-style widths) at sizes the oracle still does in seconds
    pool = [16, 20, 34, 64, 66, 100, 128, 192, 256, 256, 320, 384, 500, 512, 512, 640, 768, 1000, 1024, 1024,
            1536, 2048, 4096, 240, 480, 960, 896, 448, 720]      # (round 5: + the unequal-width stage kernels' lengths)
    limit = 48_000_000
t0, done, skipped = time.time(), 0, 0
while time.time() - t0 < budget:
    nd = int(rng.choice([2, 3, 3, 3, 4] if mode == 'small' else [2, 3, 3, 3]))
    shape = tuple(int(rng.choice(pool)) for _ in range(nd))
    if np.prod(shape) > limit or (mode == 'mid' and np.prod(shape) < 200_000):
        continue
    P = int(rng.choice([1, 2, 4, 4, 8, 8, 3, 6]))
    dt = str(rng.choice(list('dDfF')))
    kw = {}
    if rng.random() < 0.3:
        kw['collapse'] = True
    if rng.random() < 0.25 and nd >= 3:
        kw['grid'] = (-1,)
    if rng.random() < 0.25:
        kw['padding'] = [1.5] * nd
        kw['axes'] = tuple((i,) for i in range(nd))
        kw.pop('collapse', None)
    if dt in 'df' and 'padding' not in kw and rng.random() < 0.3:
        # real-to-real stages on a random suffix of single-axis groups (transforms= dict)
        kw['axes'] = tuple((i,) for i in range(nd))
        kw.pop('collapse', None)
        first = int(rng.integers(1, nd)) if nd > 1 else 0
        kw['r2r'] = {(i,): int(rng.integers(3, 11)) for i in range(first, nd)}
    os.environ['GFFT_RELAY'] = str(rng.choice(['0', '1', 'measure']))
    os.environ['GFFT_FUSE_PACK'] = str(rng.choice(['0', '1', '1']))
    os.environ['GFFT_WIRE'] = str(rng.choice(['torch', 'native', 'native']))
    pipeline.Pipeline.CHUNKS = int(rng.choice([1, 2, 3, 4]))
    try:
        cases.check_pfft_vs_oracle(P, shape, dt, seed=int(rng.integers(1 << 30)), **kw)
        done += 1
    except AssertionError as e:
        msg = str(e)
        if msg == '' or 'size' in msg or 'assert n >= size' in msg:
            skipped += 1
            continue
        print('FAIL', P, shape, dt, kw, os.environ['GFFT_RELAY'], os.environ['GFFT_FUSE_PACK'], os.environ['GFFT_WIRE'], msg[:300], flush=True)
        raise
    except RuntimeError as e:
        if 'AssertionError' in str(e) and 'n >= size' in str(e):
            skipped += 1
            continue
        print('FAIL', P, shape, dt, kw, os.environ['GFFT_RELAY'], os.environ['GFFT_FUSE_PACK'], os.environ['GFFT_WIRE'], flush=True)
        raise
print('stress seed %d: %d configurations checked, %d skipped (invalid for the rank count), %.0f s' % (seed, done, skipped, time.time() - t0))
