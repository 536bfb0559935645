# round 5, final code: the GPU suite, smoke, the headline under rocprofv3 (four passes) and plain, survey, probes of the final layout
mkdir -p gpurun_out/r05z
(timeout 2900 python -m pytest tests -m gpu -q 2>&1 | tail -8) > gpurun_out/r05z/gputests.txt; cat gpurun_out/r05z/gputests.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 > gpurun_out/r05z/smoke.txt
bash tools/prof.sh r05_bench python bench.py --steps 6 --warmup 2 --no-cpu > gpurun_out/r05z/prof_bench.log 2>&1
timeout 600 python bench.py --no-cpu > gpurun_out/r05z/bench_plain.json 2> gpurun_out/r05z/bench_plain.err
timeout 1500 python bench.py > gpurun_out/r05z/bench_plain_cpu.json 2> gpurun_out/r05z/bench_plain_cpu.err
timeout 900 python tools/survey.py > gpurun_out/r05z/survey.txt 2>&1
(timeout 300 python tools/tile_major_probe.py 2>&1 | grep -v "^/opt") > gpurun_out/r05z/tile_major_probe.txt
(timeout 600 python tools/strided_bound_probe.py 2>&1 | grep -v "^/opt") > gpurun_out/r05z/strided_bound.txt
timeout 900 python tools/stress.py 6 100 mid > gpurun_out/r05z/stress.txt 2>&1; tail -1 gpurun_out/r05z/stress.txt
timeout 600 python tools/stress_serial.py 6 80 > gpurun_out/r05z/stress_serial.txt 2>&1; tail -1 gpurun_out/r05z/stress_serial.txt
timeout 600 python tools/stress_serial.py 7 80 mid > gpurun_out/r05z/stress_serial_mid.txt 2>&1; tail -1 gpurun_out/r05z/stress_serial_mid.txt
