python tools/ab_combo_probe.py -n 1024 -d D "ws_plane_skew=0" "ws_plane_skew=16" "ws_plane_skew=48" "ws_plane_skew=272" 2>&1 | grep -v "^/opt" | tail -14
python tools/ab_combo_probe.py -n 1024 -d d "ws_plane_skew=0" "ws_plane_skew=16" "ws_plane_skew=48" 2>&1 | grep -v "^/opt" | tail -11
python tools/ab_combo_probe.py -n 1024 -d F "ws_plane_skew=0" "ws_plane_skew=32" "ws_plane_skew=96" 2>&1 | grep -v "^/opt" | tail -11
python tools/ab_combo_probe.py -n 512 -d D "ws_plane_skew=0" "ws_plane_skew=16" "ws_plane_skew=48" 2>&1 | grep -v "^/opt" | tail -11
python tools/ab_combo_probe.py -n 1024 -d f "ws_plane_skew=0" "ws_plane_skew=32" "ws_plane_skew=96" 2>&1 | grep -v "^/opt" | tail -11
