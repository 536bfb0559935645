#!/usr/bin/env python3
"""Developer tool (round 6): randomised stress of the fused slab pairs (gfft_plan_create_guru2, PFFT._fuse_pairs, pipeline._PairStage) on
one GPU with thread-ranks, against the oracle at rounding level: slab grids of 2 / 4 / 8 ranks, planes of 512^2 / 1024^2 / 512 x 1024 / 1024 x 512, complex128
and complex64, staged and pipelined wires, 1-4 chunks of planes, packed and natural exchange buffers, short hand-off rings so that
launches of 8 planes already fuse.   usage: stress_pairs.py <seed> <seconds>"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import subprocess
import numpy as np
from tests import cases
from mpi4py_fft_amd import _lib, pipeline, PFFT

_here = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'fake_rccl')
_so = os.path.join(_here, 'libfake_rccl.so')
if not os.path.exists(_so):
    subprocess.check_call(['/opt/rocm/bin/hipcc', '-O2', '-std=c++17', '-fPIC', '-shared', '-x', 'hip',
                           '--offload-arch=gfx950', os.path.join(_here, 'fake_rccl.cpp'), '-o', _so])
_lib.check_wire(_lib.lib().gfft_rccl_load(_so.encode()))
pipeline.Pipeline.MIN_CHUNK_BYTES = 0

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 120.0
rng = np.random.default_rng(seed)
t0, done, fused = time.time(), 0, 0
while time.time() - t0 < budget:
    P = int(rng.choice([2, 4, 8]))
    n, n2 = [(512, 512), (512, 512), (1024, 1024), (512, 1024), (1024, 512)][int(rng.integers(5))]      # (axis 1, axis 2) of the local planes
    dt = 'D' if (n, n2) != (1024, 1024) else str(rng.choice(['D', 'F']))
    planes = int(rng.choice([4, 8, 8, 12, 16, 16, 24]))               # per rank
    if P * planes * n * n2 > 72_000_000:
        continue
    ring = int(rng.choice([4, 6, 8]))
    _lib.set_option('fuse2_ring', ring)
    _lib.set_option('fuse2_lag', ring // 2)
    os.environ['GFFT_FUSE_PACK'] = str(rng.choice(['0', '1', '1']))
    os.environ['GFFT_WIRE'] = str(rng.choice(['torch', 'native', 'native']))
    os.environ['GFFT_FUSE_PAIRS'] = str(rng.choice(['1', '1', '1', '0']))
    pipeline.Pipeline.CHUNKS = int(rng.choice([1, 2, 3, 4]))
    shape = (P * planes, n, n2)

    def probe(comm):
        f = PFFT(comm, shape, dtype=dt, grid=[P, 1, 1])
        out = (len(f.forward._pairs), None if f.pipeline is None else f.pipeline.layout)
        f.destroy()
        return out
    how = cases.run_ranks(P, probe)[0]
    fused += bool(how[0]) or how[1] == 'slab-pair'
    try:
        cases.check_pfft_vs_oracle(P, shape, dt, seed=int(rng.integers(1 << 30)), grid=[P, 1, 1])
    except Exception:
        print('FAIL', P, shape, dt, ring, dict((k, os.environ[k]) for k in ('GFFT_FUSE_PACK', 'GFFT_WIRE', 'GFFT_FUSE_PAIRS')), pipeline.Pipeline.CHUNKS, how, flush=True)
        raise
    done += 1
_lib.set_option('fuse2_ring', 0)
_lib.set_option('fuse2_lag', 0)
print('pair stress seed %d: %d configurations checked (%d of them through a fused pair), %.0f s' % (seed, done, fused, time.time() - t0))
