cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_tracks.py tests/test_gpu_roaming_c2.py tests/test_gpu_safety.py tests/test_gpu_g10.py -x -q -m gpu > gpurun_out/r03_t19.log 2>&1; rc=$?; echo "tests rc $rc"; tail -3 gpurun_out/r03_t19.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 400 python tests/dev/soak_tracks.py 240 > gpurun_out/r03_soak19.log 2>&1; rc=$?; echo "soak rc $rc"; tail -1 gpurun_out/r03_soak19.log
