mkdir -p gpurun_out/r05g
(timeout 1500 python -m pytest tests/test_gpu_serial.py -x -q -k "3x5x2k or two_pass_235 or small_prime or lengths" 2>&1 | tail -15) > gpurun_out/r05g/tests.txt; cat gpurun_out/r05g/tests.txt
(python tools/ab_combo_probe.py -n 896 -d D "mixv=0" "mixv=1" 2>&1 | grep -v "^/opt\|AMD Radeon") > gpurun_out/r05g/ab_896D.txt; cat gpurun_out/r05g/ab_896D.txt
(python tools/ab_combo_probe.py -n 896 -d d "mixv=0" "mixv=1" 2>&1 | grep -v "^/opt\|AMD Radeon") > gpurun_out/r05g/ab_896d.txt; cat gpurun_out/r05g/ab_896d.txt
