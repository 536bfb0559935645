# round 5, first GPU batch: counters at HEAD for the odd-width far-axis pass and the C5 stages, baseline bench + survey
mkdir -p gpurun_out/r05a
bash tools/prof.sh r05_r2c_d2048 python tools/prof_cases.py r2c_d2048 > gpurun_out/r05a/prof_r2c.log 2>&1
STAGE_PROBE_ONLY=aligned bash tools/prof.sh r05_c5odd python tools/stage_probe.py c5odd > gpurun_out/r05a/prof_c5odd.log 2>&1
timeout 600 python tools/stage_probe.py c5odd > gpurun_out/r05a/stage_probe_c5odd.txt 2>&1
timeout 600 python bench.py --no-cpu > gpurun_out/r05a/bench_plain.json 2> gpurun_out/r05a/bench_plain.err
timeout 900 python tools/survey.py > gpurun_out/r05a/survey.txt 2>&1
tail -5 gpurun_out/r05a/survey.txt
