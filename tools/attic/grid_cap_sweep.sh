for cap in 256 512 1024 2048 4096 8192 16384 65536; do
  GFFT_GRID_CAP=$cap python bench.py --no-cpu --steps 8 --warmup 2 2>/dev/null | grep "^{" | python -c "
import json,sys; d=json.load(sys.stdin); print('cap $cap', d['ms_per_step'], d['roofline']['all_kernels'], d['hbm_copy_ceiling']['gbs'])"
done
