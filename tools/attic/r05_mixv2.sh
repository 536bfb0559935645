mkdir -p gpurun_out/r05g
(timeout 2400 python -m pytest tests/test_gpu_serial.py -x -q -k "3x5x2k or two_pass_235 or generated_unequal or lengths" 2>&1 | tail -15) > gpurun_out/r05g/tests2.txt; cat gpurun_out/r05g/tests2.txt
(python tools/ab_combo_probe.py -n 840 -d D "mixv=0" "mixv=1" 2>&1 | grep -v "^/opt\|AMD Radeon\|max.diff") > gpurun_out/r05g/ab_840.txt; cat gpurun_out/r05g/ab_840.txt
(python tools/ab_combo_probe.py -n 600x1000x360 -d d "mixv=0" "mixv=1" 2>&1 | grep -v "^/opt\|AMD Radeon\|max.diff") > gpurun_out/r05g/ab_600.txt; cat gpurun_out/r05g/ab_600.txt
