mkdir -p gpurun_out/r05w
{
echo "# the c2r pairs' ring loads no longer serialised by the sign applied behind each (libgfft_new.so) against the build before (libgfft_old.so): tools/ab_combo_probe.py, processes alternating"
for rnd in 1 2; do
for spec in "1024 d" "1024x1024x2048 d" "1024x512x1024 d"; do
  set -- $spec
  for lib in libgfft_old.so libgfft_new.so; do
    echo "== $lib shape $1 dtype $2"
    GFFT_AB_LIB=$lib python tools/ab_combo_probe.py -n $1 -d $2 "wtile=1" 2>&1 | grep "per step\|backward passes"
  done
done
done
} > gpurun_out/r05w/ab_c2r_sign.txt 2>&1
cat gpurun_out/r05w/ab_c2r_sign.txt
GFFT_AB_LIB=libgfft_new.so python -m pytest tests/test_gpu_fused2.py -q -x -k "real" 2>&1 | tail -3
