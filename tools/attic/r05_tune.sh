mkdir -p gpurun_out/r05t
for i in 1 2 3; do
  for t in 0 1; do
    GFFT_TUNE=$t timeout 600 python bench.py --no-cpu 2>/dev/null > gpurun_out/r05t/bench_t${t}_$i.json
    python - <<PY
import json
d=json.load(open('gpurun_out/r05t/bench_t${t}_$i.json'))
print('GFFT_TUNE=${t} run ${i}: %.3f ms per step' % d['ms_per_step'], d['roofline']['all_kernels'], d['config'].get('tuned_ws_offsets_kib'), 'copy', d['hbm_copy_ceiling']['gbs'])
PY
  done
done 2>&1 | tee gpurun_out/r05t/tune_ab.txt
