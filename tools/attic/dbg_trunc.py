import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mpi4py_fft_amd import FFT, _lib
for shp, axis in (((160, 6), 0), ((5, 640), 1)):
    f = FFT(shp, axis, dtype='D', padding=1.5)
    print(shp, f._fused_trunc, _lib.lib().gfft_last_error().decode())
    print(_lib.engine().plan_describe(f.fwd._plan))
