mkdir -p gpurun_out/r05b
(timeout 1200 python -m pytest tests/test_gpu_pipeline.py -x -q -k "admission" 2>&1 | tail -15) > gpurun_out/r05b/gates_test.txt
cat gpurun_out/r05b/gates_test.txt
(timeout 900 python -m pytest tests/test_bench_launcher.py -x -q -m gpu 2>&1 | tail -8) > gpurun_out/r05b/launcher_gpu.txt
cat gpurun_out/r05b/launcher_gpu.txt
timeout 1200 python bench.py > gpurun_out/r05b/bench_full.json 2> gpurun_out/r05b/bench_full.err
tail -3 gpurun_out/r05b/bench_full.err
python - <<'PY'
import json
d = json.load(open('gpurun_out/r05b/bench_full.json'))
print(d['ms_per_step'], d['value'], json.dumps(d['config'])[:1200])
print(json.dumps(d.get('cpu_baseline'))[:1500])
PY
