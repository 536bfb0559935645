mkdir -p gpurun_out/r04t
(timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -6) > gpurun_out/r04t/gputests.txt
cat gpurun_out/r04t/gputests.txt
for v in 17 0 15 17 0 15; do GFFT_VARIANT_COLS=$v timeout 300 python tools/variant_cols_probe.py 2>&1 | grep -v amdgpu; GFFT_VARIANT_COLS=$v timeout 300 python tools/survey.py 2>&1 | grep "C4@8"; done > gpurun_out/r04t/variant_cols.txt 2>&1
cat gpurun_out/r04t/variant_cols.txt
