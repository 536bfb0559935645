#!/bin/bash
# round 6, last third: rocprofv3 kernel stats + FETCH / WRITE / LDS counters of the kernels that arrived after the pair counters above --
# the pairs on unequal planes and the 32-line n = 512 pair (tools/unequal_pair_probe.py, ... n512, ... f32n512), the one-exchange strided tiles and
# the non-temporal row kernels (tools/cols_variant_probe.py)
mkdir -p gpurun_out
out=gpurun_out/r06_new_kernel_counters.txt
: > $out
sect() {  # tag, title
  echo "== $2" >> $out
  for f in kernel_stats pmc_fetch pmc_write pmc_lds; do echo "-- $f" >> $out; grep -h "gfft::" gpurun_out/prof_$1/$f.txt | head -40 | cut -c1-330 >> $out; done
}
bash tools/prof.sh uneq python tools/unequal_pair_probe.py > /dev/null 2>&1
sect uneq "tools/unequal_pair_probe.py under rocprofv3 (fftn of six non-cubic shapes and four slab pairs, fused and as stand-alone launches)"
bash tools/prof.sh n512 python tools/unequal_pair_probe.py n512 > /dev/null 2>&1
sect n512 "tools/unequal_pair_probe.py n512 (the square n = 512 pair on 16- and 32-line tiles)"
bash tools/prof.sh f32n512 python tools/unequal_pair_probe.py f32n512 > /dev/null 2>&1
sect f32n512 "tools/unequal_pair_probe.py f32n512 (the complex64 n = 512 pair)"
bash tools/prof.sh colsd python tools/cols_variant_probe.py D 16,0 512x512x512:1 256x256x256:1 256x1024x1024:0 2048x256x512:0 > /dev/null 2>&1
sect colsd "tools/cols_variant_probe.py D 16,0 (strided complex128: former tables against the new tiles / streams)"
bash tools/prof.sh colsf python tools/cols_variant_probe.py F 16,0 512x512x512:1 512x512x512:0 256x1024x1024:0 1024x256x1024:0 2048x512x512:0 > /dev/null 2>&1
sect colsf "tools/cols_variant_probe.py F 16,0 (strided complex64)"
PROBE_OPT=variant_rows bash tools/prof.sh rowsd python tools/cols_variant_probe.py D 16,0 256x512x1024:2 512x512x512:2 256x256x256:2 > /dev/null 2>&1
sect rowsd "PROBE_OPT=variant_rows tools/cols_variant_probe.py D 16,0 (complex128 rows)"
wc -l $out
