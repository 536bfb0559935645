mkdir -p gpurun_out/r04p
(timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -6) > gpurun_out/r04p/gputests.txt
cat gpurun_out/r04p/gputests.txt
bash tools/prof.sh r04_bench python bench.py --steps 6 --warmup 2 --no-cpu > gpurun_out/r04p/prof.log 2>&1
grep -h "LDS" gpurun_out/prof_r04_bench/pmc_lds.txt | head -8
timeout 600 python bench.py > gpurun_out/r04p/bench_plain.json 2> gpurun_out/r04p/bench_plain.err
python - <<'PY'
import json
d = json.load(open('gpurun_out/r04p/bench_plain.json'))
print(d['ms_per_step'], d['value'], json.dumps(d['roofline'])[:900])
print(json.dumps(d['whole_transform_hbm']), d['hbm_copy_ceiling']['gbs'])
PY
