# round 5 records: A/B of the workspace-pitch rule and the c2r pair of 2048-real rows (same arrays, plans alternating)
mkdir -p gpurun_out/r05r
{
echo "# tools/ab_combo_probe.py, one box, same caller arrays per shape, plan sets alternating (5 rounds x 10 steps), then per-pass times"
echo "# pitch129=0: workspace pitch = width rounded up to 128-byte lines (+256 B off multiples of 2 KiB) -- 1032 / 2064 entries = 129 x 2^k for these shapes"
echo "# pitch129=1 (default since round 5): the next pitch whose odd part is not 129 (1040 / 2080 = 65 x 2^k)"
echo "# ws_plane_skew=16: the old pitch with consecutive workspace planes 256 bytes further apart (what located the effect)"
for spec in "1024x1024x2048 d" "2048x512x2048 d" "512x1024x2048 D" "1024x1024x4096 f"; do
  set -- $spec
  echo "== shape $1 dtype $2"
  python tools/ab_combo_probe.py -n $1 -d $2 "pitch129=0,ws_plane_skew=0" "pitch129=1" "pitch129=0,ws_plane_skew=16" 2>&1 | grep -v "^/opt\|AMD Radeon"
done
echo "== the c2r pair on rows of 2048 reals (option c2r_2048), under both pitches: shape 1024x1024x2048 dtype d"
python tools/ab_combo_probe.py -n 1024x1024x2048 -d d "c2r_2048=0,pitch129=1" "c2r_2048=1" "c2r_2048=1,pitch129=0" "c2r_2048=0,pitch129=0" 2>&1 | grep -v "^/opt\|AMD Radeon"
echo "== shapes whose pitches are 17 / 33 / 65 / 257 x 2^k entries: a plane skew changes nothing (or loses)"
for spec in "1024x1024x1024 D" "1024x1024x1024 d" "1024x1024x512 d" "512x512x4096 D"; do
  set -- $spec
  echo "== shape $1 dtype $2"
  python tools/ab_combo_probe.py -n $1 -d $2 "ws_plane_skew=0" "ws_plane_skew=16" 2>&1 | grep -v "^/opt\|AMD Radeon"
done
} > gpurun_out/r05r/ab_pitch129.txt 2>&1
tail -5 gpurun_out/r05r/ab_pitch129.txt
{
echo "# tools/serial_ab_probe.py: one serial plan per option set, same arrays, alternating (5 rounds x 20 executions)"
python tools/serial_ab_probe.py 128x1048576 F 1 "fuse2_f32=0,fuse2_kinds=126" "fuse2_f32=1,fuse2_kinds=126"
python tools/serial_ab_probe.py 256x1024x1024 F 1,2 "fuse2_f32=0,fuse2_kinds=126" "fuse2_f32=1,fuse2_kinds=126"
python tools/serial_ab_probe.py 64x1048576 D 1 "fuse2=0,fuse2_kinds=126" "fuse2=1,fuse2_kinds=126"
} 2>&1 | grep -v "^/opt" > gpurun_out/r05r/c2_pairs.txt
cat gpurun_out/r05r/c2_pairs.txt

# 960^3: the two-pass plans of rounds 1-4 against the one-pass kernels, then the rocprofv3 passes of the new kernels
{
python tools/ab_combo_probe.py -n 960 -d D "mixv=0" "mixv=1"
python tools/ab_combo_probe.py -n 960 -d F "mixv=0" "mixv=1"
python tools/ab_combo_probe.py -n 960 -d d "mixv=0" "mixv=1"
python tools/ab_combo_probe.py -n 720x1200x480 -d D "mixv=0" "mixv=1"
} 2>&1 | grep -v "^/opt\|AMD Radeon" > gpurun_out/r05r/ab_mixv.txt
cat gpurun_out/r05r/ab_mixv.txt
bash tools/prof.sh r05_c960 python tools/prof_cases.py c960 c960f > gpurun_out/r05r/prof_c960.log 2>&1
bash tools/prof.sh r05_r2c_d2048 python tools/prof_cases.py r2c_d2048 > gpurun_out/r05r/prof_r2c.log 2>&1
# what bounds the stand-alone strided pass (needs mpi4py-fft_amd/libgfft_var.so: fft_pow2_f64.hip built with -DGFFT_VARIANTS)
if [ -e mpi4py-fft_amd/libgfft_var.so ]; then
  GFFT_AB_LIB=libgfft_var.so python tools/strided_bound_probe.py 2>&1 | grep -v "^/opt" > gpurun_out/r05r/strided_bound.txt; cat gpurun_out/r05r/strided_bound.txt
fi
python tools/gate_probe.py 2>&1 | grep -v "^/opt" > gpurun_out/r05r/gate_cost.txt
# the headline: four rocprofv3 passes over bench.py, then two plain runs (the second with the cpu_baseline leg at 1024^3)
bash tools/prof.sh r05_bench python bench.py --steps 6 --warmup 2 --no-cpu > gpurun_out/r05r/prof_bench.log 2>&1
timeout 600 python bench.py --no-cpu > gpurun_out/r05r/bench_plain.json 2> gpurun_out/r05r/bench_plain.err
timeout 1500 python bench.py > gpurun_out/r05r/bench_plain_cpu.json 2> gpurun_out/r05r/bench_plain_cpu.err
timeout 900 python tools/survey.py > gpurun_out/r05r/survey.txt 2>&1
STAGE_PROBE_ONLY=aligned bash tools/prof.sh r05_c5odd python tools/stage_probe.py c5odd > gpurun_out/r05r/prof_c5odd.log 2>&1
timeout 600 python tools/stage_probe.py all > gpurun_out/r05r/stage_probe.txt 2>&1
timeout 900 python tools/stress.py 5 120 mid > gpurun_out/r05r/stress.txt 2>&1; tail -3 gpurun_out/r05r/stress.txt
timeout 600 python tools/stress_serial.py 5 90 > gpurun_out/r05r/stress_serial.txt 2>&1; tail -3 gpurun_out/r05r/stress_serial.txt
