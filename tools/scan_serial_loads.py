#!/usr/bin/env python3
"""Developer aid (round 5): find kernels whose tile loads the compiler SERIALISED -- a load, `s_waitcnt vmcnt(0)`, the next load ...
-- by scanning the gfx950 ISA of the kernel tables (hipcc -S --cuda-device-only, no GPU needed).  A register-resident pass wants
all R loads of a tile in flight at once; a chain of loads each waited for costs R memory round trips per tile.  Found this way:
the c2r rows of the fused real pairs (16 round trips to the ring per tile: -13 % on the launch once fixed) and 55 unequal-width
stage kernels of fft_mixv_f64.hip (-10 ... -16 % per 3-D step on 750 / 1050 / 1260), both caused by the sign of
conjugation-on-load applied to each value as it arrives (fft_pow2_body.inc NO_SIGN / DEFER_SIGN, fft_pow2_impl.h serial_loads_f64).
usage: scan_serial_loads.py [--min-chain 4] fft_mixv_f64.hip fft_fused_real_f64.hip ...   (names relative to mpi4py-fft_amd/csrc)"""
import os, re, subprocess, sys, tempfile

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
csrc = os.path.join(root, 'mpi4py-fft_amd', 'csrc')
args = sys.argv[1:]
thr = 4
if args and args[0] == '--min-chain':
    thr = int(args[1]); args = args[2:]
for src in args:
    with tempfile.TemporaryDirectory() as tmp:
        asm = os.path.join(tmp, 'k.s')
        subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-S', '--cuda-device-only',
                               os.path.join(csrc, src), '-o', asm], stderr=subprocess.DEVNULL)
        lines = open(asm).read().split('\n')
    name, best, chain, last = None, {}, 0, -100
    for i, l in enumerate(lines):
        m = re.match(r'^(_Z\S+):', l)
        if m:
            name, chain, last = m.group(1), 0, -100
            best[name] = 0
        if name and re.search(r'\b(global_load|buffer_load)_dword', l):
            if any('s_waitcnt vmcnt(0)' in x for x in lines[i + 1:i + 4]):
                chain = chain + 1 if i - last < 10 else 1
                last = i
                best[name] = max(best[name], chain)
            else:
                chain = 0
    hits = sorted((b, k) for k, b in best.items() if b >= thr)
    print('%s: %d kernels, %d with a chain of >= %d serialised loads' % (src, len(best), len(hits), thr))
    for b, k in hits:
        m = re.search(r'I([df])Li(\d+)ELi(\d+)ELi(\d+)ELb([01])ELb[01]ELi(\d+)ELi(\d+)ELi(\d+)ELb', k)
        what = ('%s n=%s R=%s T=%s %s MINW=%s FLAGS=%s MODE=%s' % (('f64' if m.group(1) == 'd' else 'f32'), m.group(2), m.group(3), m.group(4),
                                                                 'cols' if m.group(5) == '1' else 'rows', m.group(6), m.group(7), m.group(8))) if m else k[:160]
        print('   chain %2d  %s' % (b, what))
