mkdir -p gpurun_out/r05w
{
echo "# the c2r kernels' X[N] load issued AHEAD of the row's loads (libgfft_new.so) against behind them (libgfft_old.so): tools/ab_combo_probe.py, processes alternating"
for rnd in 1 2; do
for spec in "1024 d" "1024 f" "1024x1024x2048 d" "768 d" "2048x1024x1024 f"; do
  set -- $spec
  for lib in libgfft_old.so libgfft_new.so; do
    echo "== $lib shape $1 dtype $2"
    GFFT_AB_LIB=$lib python tools/ab_combo_probe.py -n $1 -d $2 "wtile=1" 2>&1 | grep "per step\|backward passes"
  done
done
done
} > gpurun_out/r05w/ab_c2r_top.txt 2>&1
cat gpurun_out/r05w/ab_c2r_top.txt
