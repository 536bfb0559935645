mkdir -p gpurun_out/r05f
{
python tools/serial_ab_probe.py 128x1048576 F 1 "fuse2_f32=0" "fuse2_f32=1,fuse2_kinds=126" "fuse2_f32=1,fuse2_f32_lean=1" "fuse2_f32=1,fuse2_kinds=110"
python tools/serial_ab_probe.py 64x1048576 F 1 "fuse2_f32=0" "fuse2_f32=1,fuse2_kinds=126" "fuse2_f32=1,fuse2_f32_lean=1" "fuse2_f32=1,fuse2_kinds=110"
} 2>&1 | grep -v "^/opt" > gpurun_out/r05f/c2f.txt
cat gpurun_out/r05f/c2f.txt
