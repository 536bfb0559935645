mkdir -p gpurun_out/r05w
python tools/attic/r05_c2r_mirror.py 2>&1 | grep -v "^/opt" | tail -60
{
echo "# tools/ab_combo_probe.py: backward real 3-D schedules as the mirror of the forward one (option c2r_mirror), same arrays"
for spec in "1024 d" "1024x1024x2048 d" "2048x1024x1024 d" "512 d"; do
  set -- $spec
  echo "== shape $1 dtype $2"
  python tools/ab_combo_probe.py -n $1 -d $2 "c2r_mirror=0" "c2r_mirror=1" 2>&1 | grep -v "^/opt\|AMD Radeon"
done
} > gpurun_out/r05w/ab_c2r_mirror.txt 2>&1
grep "per step\|^==\|passes" gpurun_out/r05w/ab_c2r_mirror.txt
