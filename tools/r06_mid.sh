# round 6, mid-round records: GPU suite, smoke, survey, the headline plain (no CPU leg)
mkdir -p gpurun_out/r06m
(timeout 1700 python -m pytest tests -m gpu -q 2>&1 | tail -8) > gpurun_out/r06m/gputests.txt; cat gpurun_out/r06m/gputests.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 > gpurun_out/r06m/smoke.txt
timeout 600 python bench.py --no-cpu > gpurun_out/r06m/bench_plain.json 2> gpurun_out/r06m/bench_plain.err
timeout 900 python tools/survey.py > gpurun_out/r06m/survey.txt 2>&1
(timeout 300 python tools/stage_probe.py all 2>&1 | grep -v "^/opt") > gpurun_out/r06m/stage_probe.txt
