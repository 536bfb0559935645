mkdir -p gpurun_out/r05g
(timeout 2400 python -m pytest tests/test_gpu_serial.py tests/test_gpu_pfft.py -x -q -k "unequal or truncation or padded or padding" 2>&1 | tail -15) > gpurun_out/r05g/tests3.txt; cat gpurun_out/r05g/tests3.txt
python tools/attic/r05_pad960.py 2>&1 | grep -v "^/opt" > gpurun_out/r05g/pad960.txt; cat gpurun_out/r05g/pad960.txt
