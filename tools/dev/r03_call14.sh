cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout -k 10 500 python -m pytest tests/test_gpu_tracks.py tests/test_gpu_g10.py tests/test_gpu_configs.py -x -q -m gpu > gpurun_out/r03_t14.log 2>&1; rc=$?; echo "tests rc $rc"; tail -3 gpurun_out/r03_t14.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 200 python tests/dev/soak_tracks.py 100 > gpurun_out/r03_soak14.log 2>&1; rc=$?; echo "soak rc $rc"; tail -1 gpurun_out/r03_soak14.log
[ $rc -eq 0 ] || exit 1
SSRS_TRACKS_LDS_ROWS=1 timeout -k 10 200 python tests/dev/soak_tracks.py 60 > gpurun_out/r03_soak14_lr.log 2>&1; rc=$?; echo "soak (staged rows) rc $rc"; tail -1 gpurun_out/r03_soak14_lr.log
[ $rc -eq 0 ] || exit 1
for v in "" "SSRS_TRACKS_LDS_ROWS=1"; do
  echo "== ramp bench $v"
  env $v timeout -k 10 200 python bench.py --potential ramp --steps 10 --warmup 2 --cpu-seconds 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('tracks/s %.3e' % d['value'], 'ms/step %.3f' % d['ms_per_step'], d['phase_ms_per_step'], 'launches', r['launches'], 'avg launch ms %.3f' % r['avg_launch_ms'])"
done
exit 0
