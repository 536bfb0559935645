python tools/ab_combo_probe.py -n 1024x1024x2048 -d d "ws_plane_skew=0" "ws_plane_skew=8" "ws_plane_skew=16" "ws_plane_skew=48" "ws_plane_skew=272" "ws_plane_skew=1048" 2>&1 | tail -22
