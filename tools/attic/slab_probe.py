#!/usr/bin/env python3
"""Developer tool: slab-interleaved execution of the axis-2 and axis-1 passes of a 1024^3 c128
transform (does the Infinity Cache absorb the intermediate?)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mpi4py_fft_amd import fftw
from mpi4py_fft_amd.array import DeviceArray

n = 1024
A = DeviceArray((n, n, n), 'D'); W = DeviceArray((n, n, n), 'D')
torch.view_as_real(A.tensor)[:64].normal_()

def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3

print(torch.cuda.get_device_name(0))
pr = fftw.fftn(A, axes=(2,), output_array=W); pc = fftw.fftn(W, axes=(1,), output_array=W)
def whole():
    pr.execute_scaled(A, W, 1.0); pc.execute_scaled(W, W, 1.0)
print('whole arrays: rows + axis-1 in place  %.2f ms' % timed(whole))
pr.destroy(); pc.destroy()
for k in (4, 8, 16, 32, 64):
    a, w = A[:k], W[:k]
    qr = fftw.fftn(a, axes=(2,), output_array=w); qc = fftw.fftn(w, axes=(1,), output_array=w)
    def slabs():
        for s in range(0, n, k):
            qr.execute_scaled(A[s:s + k], W[s:s + k], 1.0)
            qc.execute_scaled(W[s:s + k], W[s:s + k], 1.0)
    def rows_only():
        for s in range(0, n, k):
            qr.execute_scaled(A[s:s + k], W[s:s + k], 1.0)
    sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
    evs = [torch.cuda.Event() for _ in range(n // k)]
    def two_streams():
        cur = torch.cuda.current_stream()
        sA.wait_stream(cur); sB.wait_stream(cur)
        for j, s in enumerate(range(0, n, k)):
            with torch.cuda.stream(sA):
                qr.execute_scaled(A[s:s + k], W[s:s + k], 1.0)
                evs[j].record(sA)
            with torch.cuda.stream(sB):
                sB.wait_event(evs[j])
                qc.execute_scaled(W[s:s + k], W[s:s + k], 1.0)
        cur.wait_stream(sA); cur.wait_stream(sB)
    print('   two streams (rows of slab s+1 overlap axis-1 of slab s): %.2f ms' % timed(two_streams), flush=True)
    def cols_only():
        for s in range(0, n, k):
            qc.execute_scaled(W[s:s + k], W[s:s + k], 1.0)
    print('slab of %2d planes (%4d MiB): both %.2f ms   rows only %.2f ms   axis-1 only (cold) %.2f ms' % (k, k * 16, timed(slabs), timed(rows_only), timed(cols_only)), flush=True)
    qr.destroy(); qc.destroy()
