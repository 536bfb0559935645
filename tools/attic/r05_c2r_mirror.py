import os, sys
os.environ.setdefault('GFFT_TUNE', '0')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from mpi4py_fft_amd import PFFT, comm, _lib, newDistArray
for shape in ((1024, 40, 1024), (64, 1024, 1024), (48, 1024, 2048)):
    rng = np.random.default_rng(3)
    x = rng.standard_normal(shape)
    ref = np.fft.rfftn(x)
    res = {}
    for m in (0, 1):
        _lib.set_option('c2r_mirror', m)
        _lib.set_option('fused3_min_mib', 0)
        f = PFFT(comm.COMM_SELF, shape, dtype='d')
        print(shape, 'c2r_mirror', m)
        print(f._fused_plans[1]._eng.plan_describe(f._fused_plans[1]._plan))
        vh = newDistArray(f, True)
        vh[...] = ref / x.size
        back = np.asarray(f.backward(vh)).copy()
        print('   backward max err vs input', np.abs(back - x).max())
        res[m] = back
        f.destroy()
    print('   mirror vs default max diff', np.abs(res[0] - res[1]).max())
_lib.set_option('c2r_mirror', 0)
