mkdir -p gpurun_out/r05g
(timeout 1200 python -m pytest tests/test_gpu_fused2.py -x -q -k "real_transforms or stand_alone_form" 2>&1 | tail -12) > gpurun_out/r05g/tests5.txt; cat gpurun_out/r05g/tests5.txt
{
echo "# tools/ab_combo_probe.py: the real pairs [r2c rows -> axis 1] / [axis 0 -> c2r rows] on n = 960 / 896 (option fuse2_mixv)"
python tools/ab_combo_probe.py -n 960 -d d "fuse2_mixv=0" "fuse2_mixv=1" 2>&1 | grep -v "^/opt\|AMD Radeon\|max.diff"
python tools/ab_combo_probe.py -n 896 -d d "fuse2_mixv=0" "fuse2_mixv=1" 2>&1 | grep -v "^/opt\|AMD Radeon\|max.diff"
} > gpurun_out/r05g/ab_real960.txt; cat gpurun_out/r05g/ab_real960.txt
