#!/bin/bash
# round 6: rocprofv3 kernel stats + FETCH / WRITE / LDS counters of the slab pair (tools/stage_probe.py slab) and of the C1 plane kernel
mkdir -p gpurun_out
out=gpurun_out/r06_pair_counters.txt
: > $out
bash tools/prof.sh slabpair python tools/stage_probe.py slab > /dev/null 2>&1
echo "== tools/stage_probe.py slab under rocprofv3 (C3, C4@2, C4 (8,1,1) c128, (8,1,1) c64: two stand-alone launches, the fused pair whole and in 2 / 4 chunks, both pass orders, the far stage)" >> $out
for f in kernel_stats pmc_fetch pmc_write pmc_lds; do echo "-- $f" >> $out; grep -h "gfft::" gpurun_out/prof_slabpair/$f.txt | grep -v "^[0-9]* *[0-9.]* *[0-9.]* *[0-9.]* *[0-9.]* *[0-9.]*%.*copy" | head -60 | cut -c1-330 >> $out; done
bash tools/prof.sh c1 python tools/c1_probe.py > /dev/null 2>&1
echo "== tools/c1_probe.py under rocprofv3 (64^3 / 128^3 / 32^3)" >> $out
for f in kernel_stats pmc_fetch pmc_write pmc_lds; do echo "-- $f" >> $out; grep -h "gfft::" gpurun_out/prof_c1/$f.txt | head -16 | cut -c1-300 >> $out; done
wc -l $out
