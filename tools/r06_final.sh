# round 6, final code: the GPU suite, smoke, the headline under rocprofv3 (four passes) and plain, survey, stage probes, stress
mkdir -p gpurun_out/r06z
(timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -150) > gpurun_out/r06z/gputests.txt; tail -8 gpurun_out/r06z/gputests.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -6 > gpurun_out/r06z/smoke.txt
bash tools/prof.sh r06_bench python bench.py --steps 6 --warmup 2 --no-cpu --no-configs > gpurun_out/r06z/prof_bench.log 2>&1
timeout 600 python bench.py --no-cpu > gpurun_out/r06z/bench_plain.json 2> gpurun_out/r06z/bench_plain.err
timeout 1500 python bench.py > gpurun_out/r06z/bench_plain_cpu.json 2> gpurun_out/r06z/bench_plain_cpu.err
timeout 900 python tools/survey.py > gpurun_out/r06z/survey.txt 2>&1
(timeout 600 python tools/stage_probe.py all 2>&1 | grep -v "^/opt") > gpurun_out/r06z/stage_probe.txt
(timeout 300 python tools/c1_probe.py 2>&1 | grep -v "^/opt") > gpurun_out/r06z/c1_probe.txt
timeout 900 python tools/stress.py 6 100 mid > gpurun_out/r06z/stress.txt 2>&1; tail -1 gpurun_out/r06z/stress.txt
timeout 600 python tools/stress_serial.py 6 80 > gpurun_out/r06z/stress_serial.txt 2>&1; tail -1 gpurun_out/r06z/stress_serial.txt
timeout 600 python tools/stress_serial.py 7 80 mid > gpurun_out/r06z/stress_serial_mid.txt 2>&1; tail -1 gpurun_out/r06z/stress_serial_mid.txt
