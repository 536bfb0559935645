mkdir -p gpurun_out/r05c
(python tools/ab_combo_probe.py -n 1024 -d D "pitch_extra=0" "pitch_extra=1" "pitch_extra=2" "pitch_extra=3" 2>&1 | grep -v "^/opt\|AMD Radeon\|max.diff" ) > gpurun_out/r05c/pitch_1024D.txt
cat gpurun_out/r05c/pitch_1024D.txt | head -8
(python tools/ab_combo_probe.py -n 1024 -d D "pitch_extra=0" "pitch_extra=4" "pitch_extra=6" "pitch_extra=14" 2>&1 | grep -v "^/opt\|AMD Radeon\|max.diff" ) > gpurun_out/r05c/pitch_1024D_b.txt
cat gpurun_out/r05c/pitch_1024D_b.txt | head -8
