#!/bin/bash
# round 6 records for the review items that end in a measured bound rather than a speed-up:
#  (4) C5's last stage, 513-wide against its 512-wide twin, per direction: times + FETCH / WRITE / LDS counters
#  (7) the complex64 four-step pair (128 x 2^20): kernel stats + counters
#  (3) the real fp32 3-D schedule with / without its fused pairs (option fuse2_f32 = 2), same arrays
mkdir -p gpurun_out
out=gpurun_out/r06_records.txt
: > $out
for twin in c5 c5odd; do for dir in fwd bwd; do
  STAGE_PROBE_ONLY=aligned STAGE_PROBE_DIR=$dir STAGE_PROBE_STAGE=2 bash tools/prof.sh ${twin}s2${dir} python tools/stage_probe.py $twin > /dev/null 2>&1
  echo "== (4) $twin stage 2 $dir (c5 = 512-wide rows, c5odd = 513-wide)" >> $out
  STAGE_PROBE_ONLY=aligned STAGE_PROBE_DIR=$dir STAGE_PROBE_STAGE=2 python tools/stage_probe.py $twin 2>&1 | grep "C5" >> $out
  for f in kernel_stats pmc_fetch pmc_write pmc_lds; do grep -h "gfft::" gpurun_out/prof_${twin}s2${dir}/$f.txt | head -4 | cut -c1-230 >> $out; done
done; done
echo "== (4b) the same stage under the XCD-contiguous tile order (GFFT_XCD_SWIZZLE=1)" >> $out
for twin in c5 c5odd; do GFFT_XCD_SWIZZLE=1 STAGE_PROBE_ONLY=aligned STAGE_PROBE_STAGE=2 python tools/stage_probe.py $twin 2>&1 | grep "C5" >> $out; done
echo "== (7) complex64 four-step pair, 128 x 2^20" >> $out
bash tools/prof.sh c2f python tools/prof_cases.py c2f > /dev/null 2>&1
for f in kernel_stats pmc_fetch pmc_write pmc_lds; do grep -h "gfft::" gpurun_out/prof_c2f/$f.txt | head -3 | cut -c1-260 >> $out; done
echo "== (3) 1024^3 r2c fp32: stand-alone passes (fuse2_f32=1) against the real fp32 pairs (fuse2_f32=2), same arrays" >> $out
python tools/ab_combo_probe.py -n 1024 -d f "fuse2_f32=1" "fuse2_f32=2" 2>&1 | grep -v "^/opt" >> $out
echo "== (3b) what bounds the complex64 strided passes (tools/strided_bound_probe_f32.py)" >> $out
GFFT_AB_LIB=libgfft_var.so python tools/strided_bound_probe_f32.py 2>&1 | grep -v "^/opt" >> $out
cat $out
