#!/usr/bin/env python3
"""Developer tool: compile one kernel TU with -Rpass-analysis=kernel-resource-usage and print
(kernel template args, VGPRs, scratch bytes, occupancy, LDS) -- used to A/B register pressure."""
import re, subprocess, sys
src = sys.argv[1]
res = subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-c', src,
                      '-o', '/dev/null', '-Rpass-analysis=kernel-resource-usage'], stderr=subprocess.PIPE, text=True)
blocks = re.split(r'remark: [^\n]*Function Name: ', res.stderr)[1:]
for b in blocks:
    name = b.split()[0]
    dem = subprocess.run(['/usr/bin/c++filt', name], stdout=subprocess.PIPE, text=True).stdout.strip()
    m = re.search(r'<(.*)>\(', dem)
    g = lambda k: (re.search(k + r': (\d+)', b) or [None, '?'])[1]
    print('%-64s vgpr %3s scratch %4s occ %s lds %s' % (m.group(1) if m else dem[:60], g('VGPRs'),
          g(r'ScratchSize \[bytes/lane\]'), g(r'Occupancy \[waves/SIMD\]'), g(r'LDS Size \[bytes/block\]')))
