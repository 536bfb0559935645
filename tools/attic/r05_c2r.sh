python tools/ab_combo_probe.py -n 1024x1024x2048 -d d "c2r_2048=0,pitch129=1" "c2r_2048=1" "c2r_2048=1,pitch129=0" "c2r_2048=0,pitch129=0" 2>&1 | grep -v "^/opt\|AMD Radeon" | tail -16
