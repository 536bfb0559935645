#!/usr/bin/env python3
"""Developer tool: round trips at sizes whose element counts exceed 2^31 / 2^32 (index-width check)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mpi4py_fft_amd import PFFT, comm


def run(shape, dt):
    fft = PFFT(comm.COMM_SELF, shape, dtype=dt)
    u = fft.forward.input_array.tensor
    r = torch.view_as_real(u) if u.is_complex() else u
    g = torch.Generator(device='cuda').manual_seed(7)
    step = max(1, 64 * (1 << 20) // max(1, r[0].numel()))
    for i in range(0, r.shape[0], step):
        r[i:i + step].copy_(torch.randn(r[i:i + step].shape, generator=g, device='cuda', dtype=r.dtype))
    # checksum of the input in slabs (no full clone)
    ref = [float(r[i:i + step].double().pow(2).sum()) for i in range(0, r.shape[0], step)]
    probe = r[::max(1, r.shape[0] // 7)].clone()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    uh = fft.forward()
    torch.cuda.synchronize(); t1 = time.perf_counter()
    # Parseval on the spectrum (c2c): sum |uh|^2 * N == sum |u|^2
    fft.backward()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    err = float((r[::max(1, r.shape[0] // 7)] - probe).double().pow(2).sum().sqrt() / probe.double().pow(2).sum().sqrt())
    chk = [float(r[i:i + step].double().pow(2).sum()) for i in range(0, r.shape[0], step)]
    worst = max(abs(a - b) / a for a, b in zip(ref, chk))
    print('%s %s: fwd %.1f ms bwd %.1f ms  sampled round-trip err %.2e  worst slab energy drift %.2e  mem %.0f GiB'
          % (shape, dt, (t1 - t0) * 1e3, (t2 - t1) * 1e3, err, worst, torch.cuda.max_memory_allocated() / 2**30), flush=True)
    fft.destroy()
    del fft, u, r, uh
    torch.cuda.empty_cache()


print(torch.cuda.get_device_name(0))
run((2048, 2048, 1024), 'F')      # 2^32 elements
run((2048, 2048, 2048), 'f')      # C5 on one GPU: 2^33 reals, 1025-wide half spectrum
run((1536, 1536, 1536), 'D')      # 3.6e9 elements, 54 GiB per array
