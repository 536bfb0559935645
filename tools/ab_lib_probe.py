#!/usr/bin/env python3
"""Developer aid: the fp64 survey lines under the library named by GFFT_AB_LIB (a file next to libgfft.so) --
run alternately against two builds on one box for a clean A/B of a compile-time choice."""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.argv = ['x']
from mpi4py_fft_amd import _lib
if os.environ.get('GFFT_AB_LIB'):
    _lib.LIBPATH = os.path.join(os.path.dirname(_lib.LIBPATH), os.environ['GFFT_AB_LIB'])
    # (an OLDER build may lack entry points this host declares: declare what exists, make the rest inert)
    _declare = _lib._declare
    def _tolerant(lib):
        class _Missing:
            restype = argtypes = None
        class _View:
            def __getattr__(self, name):
                try:
                    return getattr(lib, name)
                except AttributeError:
                    return _Missing()
        return _declare(_View())
    _lib._declare = _tolerant
    _lib.check_async = lambda: None
src = open(os.path.join(root, 'tools', 'survey.py')).read()
exec(src.split("print(torch.cuda.get_device_name(0))")[0])
tag = os.environ.get('GFFT_AB_LIB', 'libgfft.so')
def P(*a, **k):
    print('[%s]' % tag, end=' '); pfft_case(*a, **k)
def Q(*a):
    print('[%s]' % tag, end=' '); plan_case(*a)
P('PFFT 1024^3 c128', (1024,) * 3, 'D')
P('PFFT 512^3 c128', (512,) * 3, 'D')
P('PFFT 768^3 c128', (768,) * 3, 'D')
P('PFFT 1024^3 r2c f64', (1024,) * 3, 'd')
P('PFFT 1536x768x768 c128', (1536, 768, 768), 'D')
P('PFFT 1000^3 c128', (1000,) * 3, 'D')
P('PFFT 683^3 c128 padded 1.5', (683, 683, 683), 'D', padding=[1.5, 1.5, 1.5])
P('PFFT 384^3 c128', (384,) * 3, 'D')
Q('(256,1024,512) axis1 c128', (256, 1024, 512), 'D', (1,))
Q('(1024,256,512) axis0 c128', (1024, 256, 512), 'D', (0,))
Q('(1024,1024,1024) axis1 c128', (1024, 1024, 1024), 'D', (1,))
Q('(1024,1024,1024) axis0 c128', (1024, 1024, 1024), 'D', (0,))
P('PFFT 1024^3 c64', (1024,) * 3, 'F')
P('PFFT 1024^3 r2c f32', (1024,) * 3, 'f')
P('PFFT 512^3 r2c f64', (512,) * 3, 'd')
P('PFFT 2048x1024x1024 r2c f32', (2048, 1024, 1024), 'f')
Q('C2 batched 1-D 2^20 c128, B=64', (64, 1 << 20), 'D', (1,))
Q('(512,1024,2048) r2c f32 axis2', (512, 1024, 2048), 'f', (2,))
Q('(512,2048,513) axis1 c64', (512, 2048, 513), 'F', (1,))
Q('(2048,512,513) axis0 c64', (2048, 512, 513), 'F', (0,))
Q('(256,512,1024) axis2 c128', (256, 512, 1024), 'D', (2,))
