#!/bin/bash
# usage: tools/lds_pass.sh <tag> <command...>   -- only the LDS-counter pass of tools/prof.sh (plus kernel durations)
set -u
tag=$1; shift
root=$(pwd); out=$root/gpurun_out/prof_$tag; mkdir -p $out
export TMPDIR=/tmp
rm -rf /tmp/rp_lds
(cd $root && rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -d /tmp/rp_lds -o lds -- "$@") > $out/pmc_lds.log 2>&1
db=$(find /tmp/rp_lds -name "*.db" | head -1)
python $root/tools/rocpd_summary.py $db > $out/pmc_lds.txt 2>&1
grep "gfft::" $out/pmc_lds.txt | cut -c1-230
