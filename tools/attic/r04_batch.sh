mkdir -p gpurun_out/r04v
(timeout 2700 python -m pytest tests -m gpu -x -q 2>&1 | tail -6) > gpurun_out/r04v/gputests.txt
cat gpurun_out/r04v/gputests.txt
bash tools/prof.sh r04_bench python bench.py --steps 6 --warmup 2 --no-cpu > gpurun_out/r04v/prof.log 2>&1
timeout 600 python bench.py > gpurun_out/r04v/bench_plain.json 2> gpurun_out/r04v/bench_plain.err
python - <<'PY'
import json
d = json.load(open('gpurun_out/r04v/bench_plain.json'))
print(d['ms_per_step'], d['value'], json.dumps(d['roofline'])[:1500])
PY
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
timeout 900 python tools/survey.py > gpurun_out/r04v/survey.txt 2>&1
