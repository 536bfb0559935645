#!/usr/bin/env python3
"""Developer tool: randomised DistArray.redistribute chains on one GPU (thread ranks): random global
shapes (uneven blocks, extents below the rank count excluded as the reference excludes them), tensor
ranks 0-2, dtypes, grids, alignment walks.  Self-checking: after every redistribution each rank must
hold exactly its local_slice() of one global array."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests import cases


one = cases.check_redistribute_chain


if __name__ == '__main__':
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    budget = float(sys.argv[2]) if len(sys.argv) > 2 else 120.0
    mid = len(sys.argv) > 3 and sys.argv[3] == 'mid'
    rng = np.random.default_rng(seed)
    t0, done = time.time(), 0
    while time.time() - t0 < budget:
        done += bool(one(rng, mid))
    print('redistribute stress seed %d: %d chains checked in %.0f s' % (seed, done, time.time() - t0))
