mkdir -p gpurun_out/r05e
for pad in 0 16 32 48; do
  echo "== GFFT_T1_ROWPAD=$pad"
  GFFT_T1_ROWPAD=$pad STAGE_PROBE_ONLY=aligned python tools/stage_probe.py all 2>&1 | grep -v "^/opt\|AMD Radeon"
done > gpurun_out/r05e/rowpad.txt 2>&1
cat gpurun_out/r05e/rowpad.txt
