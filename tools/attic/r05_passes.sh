mkdir -p gpurun_out/r05w
{
for spec in "768 D" "1000 D" "1152 F" "1024x1024x2048 D" "2048x1024x1024 D" "1024 d" "768 d"; do
  set -- $spec
  echo "== shape $1 dtype $2"
  python tools/ab_combo_probe.py -n $1 -d $2 "wtile=1" 2>&1 | grep -v "^/opt\|AMD Radeon"
done
} > gpurun_out/r05w/passes.txt 2>&1
cat gpurun_out/r05w/passes.txt
