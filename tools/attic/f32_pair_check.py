import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from mpi4py_fft_amd import fftw, zeros, _lib
_lib.set_option('fuse2_f32', 1)
shape = (1024, 40, 1024)
rng = np.random.default_rng(3)
x = (rng.standard_normal(shape) + 1j * rng.standard_normal(shape)).astype('F')
a = zeros(shape, 'F'); f = fftw.fftn(a, axes=(0, 1, 2)); b = fftw.ifftn(f.output_array, axes=(0, 1, 2), output_array=zeros(shape, 'F'))
print(_lib.engine().plan_describe(f._plan))
a[...] = x
got = np.asarray(f.execute_scaled(a, f.output_array, 1.0))
ref = np.fft.fftn(x.astype('D'))
print('f32 fused fwd err', np.abs(got - ref).max() / np.abs(ref).max())
back = np.asarray(b.execute_scaled(f.output_array, b.output_array, 1.0 / x.size))
print('f32 fused round trip', np.abs(back - x).max() / np.abs(x).max())
