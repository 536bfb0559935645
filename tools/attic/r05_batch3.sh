mkdir -p gpurun_out/r05d
(timeout 2600 python -m pytest tests -m gpu -q 2>&1 | tail -25) > gpurun_out/r05d/gputests.txt; cat gpurun_out/r05d/gputests.txt
bash tools/r05_rowpad.sh
