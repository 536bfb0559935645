mkdir -p gpurun_out/r05w
python tools/attic/r05_wtile.py 2>&1 | grep -v "^/opt" | tail -24
(python tools/ab_combo_probe.py -n 1024 -d D "wtile=0" "wtile=1" 2>&1 | grep -v "^/opt\|AMD Radeon") > gpurun_out/r05w/ab_wtile_1024D.txt; cat gpurun_out/r05w/ab_wtile_1024D.txt
