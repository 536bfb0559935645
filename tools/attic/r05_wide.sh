mkdir -p gpurun_out/r05g
for spec in "960 D" "896 D" "840 D" "672 D" "480 D" "960 d" "1440x480x1440 D"; do
  set -- $spec
  echo "== $1 $2"
  python tools/ab_combo_probe.py -n $1 -d $2 "mixv_wide=0" "mixv_wide=1" 2>&1 | grep -v "^/opt\|AMD Radeon\|max.diff" 
done > gpurun_out/r05g/ab_wide.txt 2>&1
cat gpurun_out/r05g/ab_wide.txt
