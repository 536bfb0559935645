mkdir -p gpurun_out/r05w
{
echo "# tools/ab_combo_probe.py: unfused complex 3-D schedules on the tile-major layout W[i0][tile][k1][.] (wtile_all: 1 backward, 2 also forward with axis 1 first)"
for spec in "768 D" "1152 F" "1536x768x768 D" "2048 F" "1024x1024x2048 D" "2048x1024x1024 D"; do
  set -- $spec
  echo "== shape $1 dtype $2"
  python tools/ab_combo_probe.py -n $1 -d $2 "wtile_all=0" "wtile_all=1" "wtile_all=2" 2>&1 | grep -v "^/opt\|AMD Radeon"
done
} > gpurun_out/r05w/ab_wtile_all.txt 2>&1
grep "per step\|^==\|max.diff\|passes\|rror" gpurun_out/r05w/ab_wtile_all.txt
