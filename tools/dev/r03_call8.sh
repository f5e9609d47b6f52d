cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_tracks.py tests/test_gpu_roaming_c2.py tests/test_gpu_g10.py -x -q -m gpu > gpurun_out/r03_t8.log 2>&1; rc=$?; echo "tests rc $rc"; tail -5 gpurun_out/r03_t8.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 200 python tests/dev/soak_tracks.py 90 > gpurun_out/r03_soak8.log 2>&1; rc=$?; echo "soak rc $rc"; tail -1 gpurun_out/r03_soak8.log
[ $rc -eq 0 ] || exit 1
: > gpurun_out/r03_diag8.txt
for v in "SSRS_TRACKS_DEBUG_ROAM=1" "SSRS_TRACKS_NO_FINE_TABLE=1" "SSRS_TRACKS_ROAM_STEPS=16384" "SSRS_TRACKS_DEAL_ROUND_ROBIN=1"; do
  echo "== $v" >> gpurun_out/r03_diag8.txt
  env $v timeout -k 10 200 python tools/dev/r03_diag.py 100000 1500000 2>&1 | grep "^pass\|^\[roam\]" | cut -c1-330 | tail -8 >> gpurun_out/r03_diag8.txt
done
cat gpurun_out/r03_diag8.txt
