#!/usr/bin/env python3
"""Developer probe: the fused pairs on planes of 512 x 1024 / 1024 x 512 points (option fuse2_mixed) against the same plans
as stand-alone launches -- plans alternating on the SAME arrays, 5 rounds x 10 executions.
  one-rank 3-D transforms of non-cubic arrays (pair = [axis 0 -> rows of axis 2]) and the local pair of a slab
  (gfft_plan_create_guru2, blocks of the strided axis on the buffer side: forward out, backward in).
usage: unequal_pair_probe.py            the unequal pairs (option fuse2_mixed 0 / 1)
       unequal_pair_probe.py n512       the square complex128 n = 512 pair on 16- against 32-line tiles (fuse2_n512 1 / 2)
       unequal_pair_probe.py f32n512    the complex64 n = 512 pair against two launches (fuse2_f32_n512 0 / 1)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mpi4py_fft_amd import _lib

eng = _lib.engine()
print(torch.cuda.get_device_name(0))


def timed(run, reps=10):
    run()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        run()
    e.record(); e.synchronize()
    return s.elapsed_time(e) / reps


OPT, VALS, REST = 'fuse2_mixed', (0, 1), 1


def ab(label, make, a, b, alg_bytes):
    plans = {}
    for v in VALS:
        _lib.set_option(OPT, v)
        plans[v] = make()
    _lib.set_option(OPT, REST)
    tot = {v: [] for v in VALS}
    for rnd in range(5):
        for v in VALS:
            tot[v].append(timed(lambda: eng.execute_ptr(plans[v], a.data_ptr(), b.data_ptr(), 1.0)))
    torch.cuda.synchronize()
    _lib.check_async()
    m = {v: sum(tot[v]) / 5 for v in tot}
    v0, v1 = VALS
    print('%-58s launches %d -> %d   %.3f -> %.3f ms (%+.1f %%)   %.2f -> %.2f of 8 TB/s' % (
        label, eng.plan_cost(plans[v0])[2], eng.plan_cost(plans[v1])[2], m[v0], m[v1], 100 * (m[v1] / m[v0] - 1),
        alg_bytes / m[v0] / 8e9, alg_bytes / m[v1] / 8e9), flush=True)
    for h in plans.values():
        eng.plan_destroy(h)


PREC, CDT, ISZ = 8, torch.complex128, 16
if len(sys.argv) > 1 and sys.argv[1] == 'f32n512':
    # the complex64 n = 512 pair (option fuse2_f32_n512) against two launches
    OPT, VALS, REST = 'fuse2_f32_n512', (0, 1), 1
    PREC, CDT, ISZ = 4, torch.complex64, 8
    SHAPES = [(512, 512, 512), (1024, 512, 512)]
    SLABS = [(256, 512, 512, 2), (512, 512, 512, 4), (1024, 512, 512, 8)]
elif len(sys.argv) > 1 and sys.argv[1] == 'n512':
    # the SQUARE n = 512 pair: 16 lines per tile on 256 threads, two workgroups per CU (option fuse2_n512 = 1) against 32 lines per
    # tile on 512 threads, one per CU (= 2, the tile shape of the unequal pairs)
    OPT, VALS, REST = 'fuse2_n512', (1, 2), 2
    SHAPES = [(512, 512, 512), (256, 512, 512)]
    SLABS = [(256, 512, 512, 2), (64, 512, 512, 8), (512, 512, 512, 4)]
else:
    SHAPES = [(512, 512, 1024), (512, 1024, 1024), (1024, 512, 512), (1024, 1024, 512), (512, 1024, 512), (1024, 512, 1024)]
    SLABS = [(256, 512, 1024, 2), (128, 512, 1024, 8), (256, 1024, 512, 2), (128, 1024, 512, 8)]

for shape in SHAPES:
    a = torch.empty(shape, dtype=CDT, device='cuda')
    torch.view_as_real(a).normal_()
    b = torch.empty_like(a)
    nbytes = a.numel() * ISZ
    for kind, name in ((-1, 'fwd'), (+1, 'bwd')):
        ab('fftn %s %s %s' % (shape, 'c128' if PREC == 8 else 'c64', name), lambda: eng.plan_create(list(shape), list(shape), [0, 1, 2], kind, PREC), a, b, 6 * nbytes)
    del a, b

for planes, n1, n2, p in SLABS:
    a = torch.empty((planes, n1, n2), dtype=CDT, device='cuda')
    torch.view_as_real(a).normal_()
    E = (n1 // p) * n2
    buf = torch.empty((p * planes * E,), dtype=CDT, device='cuda')
    nbytes = a.numel() * ISZ
    ab('slab pair (%d,%d,%d) p=%d fwd' % (planes, n1, n2, p),
       lambda: eng.plan_create_guru2(PREC, -1, (n1, n2, n2), (n2, 1, 1), (planes, n1 * n2, E), True, 1, 0, p, planes * E), a, buf, 4 * nbytes)
    ab('slab pair (%d,%d,%d) p=%d bwd' % (planes, n1, n2, p),
       lambda: eng.plan_create_guru2(PREC, +1, (n1, n2, n2), (n2, 1, 1), (planes, E, n1 * n2), True, p, planes * E, 1, 0), buf, a, 4 * nbytes)
    del a, buf
