mkdir -p gpurun_out/r05g
(timeout 1200 python -m pytest tests/test_gpu_fused2.py -x -q -k "unequal" 2>&1 | tail -12) > gpurun_out/r05g/tests4.txt; cat gpurun_out/r05g/tests4.txt
(python tools/ab_combo_probe.py -n 960 -d D "fuse2_mixv=0" "fuse2_mixv=1" 2>&1 | grep -v "^/opt\|AMD Radeon\|max.diff") > gpurun_out/r05g/ab_f960.txt; cat gpurun_out/r05g/ab_f960.txt
(python tools/ab_combo_probe.py -n 896 -d D "fuse2_mixv=0" "fuse2_mixv=1" 2>&1 | grep -v "^/opt\|AMD Radeon\|max.diff") > gpurun_out/r05g/ab_f896.txt; cat gpurun_out/r05g/ab_f896.txt
