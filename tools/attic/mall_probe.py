#!/usr/bin/env python3
"""Developer tool: does the 256 MiB Infinity Cache absorb a producer->consumer hand-off?
Two dependent streaming passes (B = f(A); C = g(B)) over 8 GiB, run whole-array vs slab by slab."""
import torch, time
n = 1 << 29          # 512 Mi complex128 = 8 GiB
A = torch.empty(n, dtype=torch.complex128, device='cuda'); torch.view_as_real(A).normal_()
B = torch.empty_like(A); C = torch.empty_like(A)

def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3

def whole():
    B.copy_(A); C.copy_(B)
def inplace_whole():
    B.copy_(A); B.mul_(1.0000001)
print(torch.cuda.get_device_name(0))
print('whole: copy+copy %.2f ms, copy+inplace-scale %.2f ms' % (timed(whole), timed(inplace_whole)))
for mib in (16, 32, 64, 128, 256, 512):
    m = mib * (1 << 20) // 16
    def slabs():
        for s in range(0, n, m):
            B[s:s + m].copy_(A[s:s + m]); C[s:s + m].copy_(B[s:s + m])
    def slabs_inplace():
        for s in range(0, n, m):
            B[s:s + m].copy_(A[s:s + m]); B[s:s + m].mul_(1.0000001)
    print('slab %4d MiB: copy+copy %.2f ms, copy+inplace-scale %.2f ms' % (mib, timed(slabs), timed(slabs_inplace)))
