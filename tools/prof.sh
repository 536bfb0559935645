#!/bin/bash
# usage: tools/prof.sh <tag> <command...>      (on the GPU box, from the repo root)
# Four rocprofv3 passes over the same command -- kernel trace + stats, then FETCH_SIZE, WRITE_SIZE and
# the LDS counters each in their own pass (PMC passes carry --kernel-trace only) -- condensed by
# tools/rocpd_summary.py into gpurun_out/prof_<tag>/{kernel_stats,pmc_fetch,pmc_write,pmc_lds}.txt
set -u
tag=$1; shift
root=$(pwd)
out=$root/gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
run() {  # name, extra rocprofv3 args...
  local name=$1; shift
  rm -rf /tmp/rp_$name
  (cd $root && rocprofv3 "$@" -d /tmp/rp_$name -o $name -- "${CMD[@]}") > $out/$name.log 2>&1
  local db=$(find /tmp/rp_$name -name "*.db" | head -1)
  if [ -n "$db" ]; then python $root/tools/rocpd_summary.py $db > $out/$name.txt 2>&1; else echo "no db for $name" > $out/$name.txt; fi
}
CMD=("$@")
run kernel_stats --kernel-trace --stats
run pmc_fetch --pmc FETCH_SIZE --kernel-trace
run pmc_write --pmc WRITE_SIZE --kernel-trace
run pmc_lds --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace
grep -h "gfft::" $out/kernel_stats.txt | head -12
grep -h "gfft::" $out/pmc_fetch.txt $out/pmc_write.txt $out/pmc_lds.txt | grep -v "^[0-9]" | head -40
