mkdir -p gpurun_out/r05g
{
for spec in "960 D" "896 D" "840 D" "672 D" "1440x480x1440 D"; do
  set -- $spec
  echo "== $1 $2  (mixv_variant 1 = fp64 strided kernels on 32 values per thread / 512 threads; fuse2 5 = the 3-D pair likewise)"
  python tools/ab_combo_probe.py -n $1 -d $2 "mixv_variant=0,fuse2=1" "mixv_variant=1,fuse2=1" "mixv_variant=1,fuse2=5" 2>&1 | grep -v "^/opt\|AMD Radeon\|max.diff"
done
for spec in "960 F" "896 F" "672 F" "960 f"; do
  set -- $spec
  echo "== $1 $2  (mixv_variant 2 = fp32 strided kernels on 16 values per thread, 128-byte segments, no spills)"
  python tools/ab_combo_probe.py -n $1 -d $2 "mixv_variant=0" "mixv_variant=2" 2>&1 | grep -v "^/opt\|AMD Radeon\|max.diff"
done
} > gpurun_out/r05g/ab_variants.txt 2>&1
cat gpurun_out/r05g/ab_variants.txt
