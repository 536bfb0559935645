#!/usr/bin/env python3
"""Developer aid: where a workgroup of the fused launch spends a tile.  Needs the trace build
(`make -C mpi4py-fft_amd/csrc trace` -> libgfft_trace.so, GFFT_FUSE2_TRACE): every workgroup stamps the
100 MHz wall clock at [0] ticket known, [1] its counter there, [2] loads arrived, [3] tile computed and stores
issued, [4] counter raised (A: after the write-through acknowledgements) for its first 96 tickets.
Prints mean / median microseconds per phase and kind of tile for the 1024^3 complex128 forward transform
(argument `c2`: 64 x 2^20)."""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
out = os.path.join(root, 'gpurun_out', 'fuse2_trace.bin')
os.makedirs(os.path.dirname(out), exist_ok=True)
os.environ['GFFT_FUSE2_DEBUG'] = '1'
os.environ['GFFT_FUSE2_TRACE_FILE'] = out
import numpy as np, torch
from mpi4py_fft_amd import PFFT, comm, _lib, fftw, zeros
_lib.LIBPATH = os.path.join(os.path.dirname(_lib.LIBPATH), 'libgfft_trace.so')
if 'r2c' in sys.argv[1:] or 'c2r' in sys.argv[1:]:
    fft = PFFT(comm.COMM_SELF, (1024, 1024, 2048) if 'wide' in sys.argv[1:] else (1024,) * 3, dtype='d')
    fft.forward.input_array.tensor.normal_()
    fft.forward()
    run = (lambda: fft.forward()) if 'r2c' in sys.argv[1:] else (lambda: fft.backward())
elif 'c2' in sys.argv[1:]:
    a = zeros((64, 1 << 20), 'D'); torch.view_as_real(a.tensor).normal_()
    p = fftw.fftn(a, axes=(1,))
    run = lambda: p.execute_scaled(a, p.output_array, 1.0)
else:
    fft = PFFT(comm.COMM_SELF, (1024,) * 3, dtype='D')
    torch.view_as_real(fft.forward.input_array.tensor).normal_()
    run = lambda: fft.forward()
for _ in range(3):
    run()
torch.cuda.synchronize()
tr = np.fromfile(out, dtype=np.uint64).reshape(1024, 96, 16).astype(np.int64)
lo, hi = 8, 95
# the default kernels (32 values per thread) have two butterfly stages and one exchange; GFFT_FUSE2=3 (the round-3
# kernels, 16 values per thread) three stages and two exchanges
three = os.environ.get('GFFT_FUSE2', '1') == '3'
if 'r2c' in sys.argv[1:] or 'c2r' in sys.argv[1:]:
    # (row and strided tiles of the real pairs have different stage counts: coarse phases only)
    names = ['wait for counter', 'loads', 'butterflies + exchanges (+ Hermitian pass)', 'issue stores', 'raise counter (A: acks; B: slowest wave)', 'to next ticket']
    pick = lambda s, nxt: [s[0], s[1], s[2], s[5], s[3], s[4], nxt]
elif three:
    names = ['wait for counter', 'loads', 'stage 1 butterflies', 'exchange 1', 'stage 2 (twiddles + butterflies)', 'exchange 2',
             'stage 3', 'issue stores', 'raise counter (A: acks)', 'to next ticket']
    pick = lambda s, nxt: [s[0], s[1], s[2], s[8], s[9], s[10], s[11], s[12], s[3], s[4], nxt]
else:
    names = ['wait for counter', 'loads', 'stage 1 butterflies', 'exchange', 'stage 2 (twiddles + butterflies)',
             'issue stores', 'raise counter (A: acks; B: slowest wave)', 'to next ticket']
    pick = lambda s, nxt: [s[0], s[1], s[2], s[8], s[9], s[10], s[3], s[4], nxt]
for kind, label in ((1, 'A tiles (producer: -> ring)'), (0, 'B tiles (consumer: ring ->)')):
    rows = []
    for b in range(1024):
        for it in range(lo, hi):
            if tr[b, it, 7] == 0 or tr[b, it + 1, 0] == 0 or (tr[b, it, 7] & 1) != kind:
                continue
            pts = pick(tr[b, it], tr[b, it + 1, 0])
            rows.append([pts[i + 1] - pts[i] for i in range(len(pts) - 1)] + [pts[-1] - pts[0]])
    r = np.array(rows, dtype=np.float64) / 100.0
    print('%s: %d samples, %.2f us per tile (median %.2f)' % (label, len(r), r[:, -1].mean(), np.median(r[:, -1])))
    for i, n in enumerate(names):
        print('   %-44s mean %6.2f  median %6.2f  p90 %6.2f us' % (n, r[:, i].mean(), np.median(r[:, i]), np.percentile(r[:, i], 90)))
