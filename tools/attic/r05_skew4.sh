for spec in "1024x1024x2048 d" "2048x512x2048 d" "512x1024x2048 D" "1024x1024x4096 f"; do
  set -- $spec
  echo "== $1 $2"
  python tools/ab_combo_probe.py -n $1 -d $2 "pitch129=0,ws_plane_skew=0" "pitch129=1" "pitch129=0,ws_plane_skew=16" 2>&1 | grep -v "^/opt\|AMD Radeon" | tail -11
done
