mkdir -p gpurun_out/r05w
{
echo "# C2 four-step pairs: time per plane against the number of planes (the ramp of a short launch), tools/serial_ab_probe.py"
for b in 32 64 128 256 512; do python tools/serial_ab_probe.py ${b}x1048576 D 1 "fuse2=1,fuse2_kinds=126" 2>&1 | grep "mean\|fused pair"; done
for b in 64 128 256 512 1024; do python tools/serial_ab_probe.py ${b}x1048576 F 1 "fuse2_f32=1,fuse2_kinds=126" 2>&1 | grep "mean\|fused pair"; done
echo "# ring / lag on the short launches"
python tools/serial_ab_probe.py 64x1048576 D 1 "fuse2_ring=0,fuse2_lag=0" "fuse2_ring=12,fuse2_lag=4" "fuse2_ring=12,fuse2_lag=3" "fuse2_ring=16,fuse2_lag=8" "fuse2_ring=16,fuse2_lag=5" "fuse2_ring=10,fuse2_lag=4" 2>&1 | grep "mean\|fused pair"
python tools/serial_ab_probe.py 128x1048576 F 1 "fuse2_ring=0,fuse2_lag=0" "fuse2_ring=24,fuse2_lag=8" "fuse2_ring=24,fuse2_lag=6" "fuse2_ring=32,fuse2_lag=16" "fuse2_ring=32,fuse2_lag=10" "fuse2_ring=16,fuse2_lag=8" "fuse2_ring=16,fuse2_lag=6" 2>&1 | grep "mean\|fused pair"
} > gpurun_out/r05w/c2_ramp.txt 2>&1
cat gpurun_out/r05w/c2_ramp.txt
