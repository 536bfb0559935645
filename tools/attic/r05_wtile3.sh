mkdir -p gpurun_out/r05w
{
echo "# tools/ab_combo_probe.py: rows of padding after every tile of the tile-major workspace (wtile_pad), same arrays"
for spec in "1024 D" "896 D" "1024x2048x1024 D" "1024 F"; do
  set -- $spec
  echo "== shape $1 dtype $2"
  python tools/ab_combo_probe.py -n $1 -d $2 "wtile=0" "wtile=1" "wtile=1,wtile_pad=1" "wtile=1,wtile_pad=3" "wtile=1,wtile_pad=8" 2>&1 | grep -v "^/opt\|AMD Radeon"
done
} > gpurun_out/r05w/ab_wtile_pad.txt 2>&1
grep "per step\|^==\|max.diff\|passes" gpurun_out/r05w/ab_wtile_pad.txt
