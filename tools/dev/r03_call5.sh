cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for v in "SSRS_TRACKS_ROAM_STEPS=16384" "SSRS_TRACKS_ROAM_STEPS=65536"; do
  echo "== $v"
  env SSRS_TRACKS_DEBUG_ROAM=1 $v timeout -k 10 200 python tools/dev/r03_diag.py 100000 800000 2>&1 | grep "^pass 0\|^\[roam\]" | head -40
done > gpurun_out/r03_diag5.txt 2>&1
cat gpurun_out/r03_diag5.txt | cut -c1-260
