import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from mpi4py_fft_amd import fftw, zeros, _lib
_lib.set_option('wtile', 1)
for shape in ((1024, 40, 1024), (1024, 64, 1024)):
    rng = np.random.default_rng(1)
    x = rng.standard_normal(shape) + 1j * rng.standard_normal(shape)
    ref = np.fft.fftn(x)
    a = zeros(shape, 'D')
    f = fftw.fftn(a, axes=(0, 1, 2))
    b = fftw.ifftn(f.output_array, axes=(0, 1, 2), output_array=zeros(shape, 'D'))
    print(_lib.engine().plan_describe(f._plan))
    a[...] = x
    got = np.asarray(f.execute_scaled(a, f.output_array, 1.0))
    print('wtile fused forward rel err', np.abs(got - ref).max() / np.abs(ref).max())
    back = np.asarray(b.execute_scaled(f.output_array, b.output_array, 1.0 / x.size))
    print('wtile fused backward err', np.abs(back - x).max())
    # void the launch: the stand-alone forms on the tile-major workspace
    _lib.set_option('fuse2_wait_ms', 0)
    f.execute_scaled(a, f.output_array, 1.0); torch.cuda.synchronize()
    try:
        np.asarray(f.output_array)
    except RuntimeError as e:
        print('voided:', str(e)[:60])
    _lib.set_option('fuse2_wait_ms', 2000)
    got = np.asarray(f.execute_scaled(a, f.output_array, 1.0))
    print('stand-alone forms after the void: rel err', np.abs(got - ref).max() / np.abs(ref).max(), 'fused pair' in _lib.engine().plan_describe(f._plan))
    f.destroy(); b.destroy()
_lib.set_option('wtile', 0)
