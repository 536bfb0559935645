#!/bin/bash
# round 6: the per-rank stages of C4 on 8 GPUs, forward against backward -- times under grid caps and chunk counts, then the
# FETCH / WRITE / LDS counters of stages 1 and 2 per direction (one direction per process: both run the same kernel)
mkdir -p gpurun_out
out=gpurun_out/r06_c4_bwd.txt
: > $out
for cap in 0 16384; do
  echo "== GFFT_GRID_CAP=$cap: stage_probe c4 (aligned)" >> $out
  GFFT_GRID_CAP=$cap STAGE_PROBE_ONLY=aligned python tools/stage_probe.py c4 2>&1 | grep C4 >> $out
done
echo "== chunk counts (K0, K1)" >> $out
STAGE_PROBE_ONLY=aligned python tools/stage_probe.py chunks 2>&1 | grep "C4@8" >> $out
for st in 1 2; do for dir in fwd bwd; do
  STAGE_PROBE_ONLY=aligned STAGE_PROBE_DIR=$dir STAGE_PROBE_STAGE=$st bash tools/prof.sh c4s${st}${dir} python tools/stage_probe.py c4 > /dev/null 2>&1
  echo "== counters: stage $st $dir" >> $out
  for f in kernel_stats pmc_fetch pmc_write pmc_lds; do grep -h "gfft::" gpurun_out/prof_c4s${st}${dir}/$f.txt | head -3 | cut -c1-260 >> $out; done
done; done
cat $out
