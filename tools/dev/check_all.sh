set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/tall.log 2>&1 || { tail -40 gpurun_out/tall.log; exit 1; }
tail -3 gpurun_out/tall.log
