import os, sys
sys.path.insert(0, os.getcwd())
src = open('tools/survey.py').read()
exec(src.split("print(torch.cuda.get_device_name(0))")[0])
print('variant_cols', os.environ.get('GFFT_VARIANT_COLS'))
plan_case('C5@8 local: (512,2048,513) axis1 c64', (512, 2048, 513), 'F', (1,))
plan_case('C5@8 local: (2048,512,513) axis0 c64', (2048, 512, 513), 'F', (0,))
plan_case('(512,2048,512) axis1 c64', (512, 2048, 512), 'F', (1,))
plan_case('(1024,1024,1024) axis1 c64', (1024, 1024, 1024), 'F', (1,))
plan_case('(1024,1024,1024) axis0 c64', (1024, 1024, 1024), 'F', (0,))
import numpy as np
from mpi4py_fft_amd import fftw, zeros
rng = np.random.default_rng(1)
for shape, ax in (((8, 2048, 48), 1), ((1024, 6, 64), 0)):
    x = (rng.standard_normal(shape) + 1j * rng.standard_normal(shape)).astype('F')
    a = zeros(shape, 'F'); f = fftw.fftn(a, axes=(ax,)); a[...] = x
    got = np.asarray(f.execute_scaled(a, f.output_array, 1.0)); ref = np.fft.fft(x.astype('D'), axis=ax)
    print('check', shape, ax, 'rel err %.2e' % (np.abs(got - ref).max() / np.abs(ref).max()))
