for spec in "1024x1024x512 d" "1024x1024x256 d" "512x512x4096 d" "2048x512x2048 d" "1024x1024x512 D" "512x1024x2048 D" "1024x512x2048 D" "512x512x4096 D" "1024x1024x2048 f" "1024x1024x4096 f" "2048x2048x256 D"; do
  set -- $spec
  echo "== $1 $2"
  python tools/ab_combo_probe.py -n $1 -d $2 "ws_plane_skew=0" "ws_plane_skew=16" 2>&1 | grep -v "^/opt\|max|diff|\|AMD Radeon" | tail -6
done
