mkdir -p gpurun_out/r05w
{
echo "# unequal-width stage kernels: conjugation-on-load applied once BEHIND the loads (libgfft_new.so) against on each value as it arrives (libgfft_old.so,"
echo "# where the compiler waited for every load before issuing the next in 55 of the fp64 kernels); and the c2r pairs' ring loads likewise.  tools/ab_combo_probe.py, processes alternating"
for rnd in 1 2; do
for spec in "840 D" "1050 D" "750 D" "1260 D" "960 D" "720x1200x480 D" "840 d" "1050 d" "960 d" "1024 d"; do
  set -- $spec
  for lib in libgfft_old.so libgfft_new.so; do
    echo "== $lib shape $1 dtype $2"
    GFFT_AB_LIB=$lib python tools/ab_combo_probe.py -n $1 -d $2 "wtile=1" 2>&1 | grep "per step\|passes"
  done
done
done
} > gpurun_out/r05w/ab_defer_sign.txt 2>&1
grep "==\|per step" gpurun_out/r05w/ab_defer_sign.txt
