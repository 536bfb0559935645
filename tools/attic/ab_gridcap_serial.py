#!/usr/bin/env python3
"""Developer probe: workgroups per launch (grid_cap, read at launch time) for the serial stage plans of the
multi-GPU configurations -- one plan, same arrays, caps alternated, 5 rounds x 10 executions."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mpi4py_fft_amd import _lib
from mpi4py_fft_amd.libfft import FFT

print(torch.cuda.get_device_name(0))
CAPS = (4096, 8192, 16384, 32768)
for shape, dt, axes in (((256, 512, 1024), 'D', (2,)), ((256, 1024, 512), 'D', (1,)), ((1024, 256, 512), 'D', (0,)),
                        ((512, 1024, 2048), 'f', (2,)), ((512, 2048, 513), 'F', (1,)), ((2048, 512, 513), 'F', (0,)),
                        ((512, 512, 512), 'D', (1,)), ((64, 1 << 20), 'D', (1,))):
    f = FFT(shape, axes=axes, dtype=dt)
    t = f.forward.input_array.tensor
    (torch.view_as_real(t) if t.is_complex() else t).normal_()
    res = {c: [] for c in CAPS}
    for rnd in range(5):
        for c in CAPS:
            _lib.set_option('grid_cap', c)
            f.forward()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(10):
                f.forward()
            e.record(); e.synchronize()
            res[c].append(s.elapsed_time(e) / 10)
    _lib.set_option('grid_cap', 0)
    print('%-18s %s axes %s: ' % (shape, dt, axes) + ' | '.join('%d: %.4f' % (c, sum(v) / len(v)) for c, v in res.items()), flush=True)
    f.destroy()
