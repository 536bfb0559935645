mkdir -p gpurun_out/r05w
{
echo "# tools/ab_combo_probe.py: tile-major workspace under the complex 3-D pair schedule (option wtile), same arrays, plans alternating"
for spec in "1024 D" "1024 F" "512 D" "960 D" "896 D" "1024x512x1024 D" "1024x2048x1024 D"; do
  set -- $spec
  echo "== shape $1 dtype $2"
  python tools/ab_combo_probe.py -n $1 -d $2 "wtile=0" "wtile=1" 2>&1 | grep -v "^/opt\|AMD Radeon"
done
} > gpurun_out/r05w/ab_wtile.txt 2>&1
grep "per step\|^==\|max.diff" gpurun_out/r05w/ab_wtile.txt
