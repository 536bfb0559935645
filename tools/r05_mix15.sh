mkdir -p gpurun_out/r05g
(timeout 1500 python -m pytest tests/test_gpu_serial.py -x -q -k "3x5x2k or two_pass_235" 2>&1 | tail -15) > gpurun_out/r05g/tests.txt; cat gpurun_out/r05g/tests.txt
(python tools/ab_combo_probe.py -n 960 -d d "mix15=0" "mix15=1" 2>&1 | grep -v "^/opt\|AMD Radeon") > gpurun_out/r05g/ab_960d.txt; cat gpurun_out/r05g/ab_960d.txt
(python tools/ab_combo_probe.py -n 960 -d f "mix15=0" "mix15=1" 2>&1 | grep -v "^/opt\|AMD Radeon") > gpurun_out/r05g/ab_960f.txt; cat gpurun_out/r05g/ab_960f.txt
