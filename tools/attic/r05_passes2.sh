mkdir -p gpurun_out/r05w
{
for spec in "1024 f" "2048x1024x1024 f" "1024 F" "512 D" "512 d" "512 f"; do
  set -- $spec
  echo "== shape $1 dtype $2"
  python tools/ab_combo_probe.py -n $1 -d $2 "wtile=1" 2>&1 | grep -v "^/opt\|AMD Radeon"
done
} > gpurun_out/r05w/passes2.txt 2>&1
cat gpurun_out/r05w/passes2.txt
