cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_tracks.py tests/test_gpu_g10.py tests/test_gpu_roaming_c2.py -x -q -m gpu > gpurun_out/r03_t11.log 2>&1; rc=$?; echo "tests rc $rc"; tail -3 gpurun_out/r03_t11.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 200 python tests/dev/soak_tracks.py 100 > gpurun_out/r03_soak11.log 2>&1; rc=$?; echo "soak rc $rc"; tail -1 gpurun_out/r03_soak11.log
[ $rc -eq 0 ] || exit 1
for v in "SSRS_TRACKS_DEBUG_ROAM=1" "" "SSRS_TRACKS_NO_LDS_ROWS=1" "SSRS_TRACKS_NO_LDS_ROWS=1 SSRS_TRACKS_NO_CHEAP_EXACT=1"; do
  echo "== ramp bench $v"
  env $v timeout -k 10 200 python bench.py --potential ramp --steps 6 --warmup 2 --cpu-seconds 0 --no-chain-probe 2> gpurun_out/r03_b11.err | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('tracks/s %.3e' % d['value'], 'ms/step %.3f' % d['ms_per_step'], d['phase_ms_per_step'], 'launches', r['launches'], 'avg launch ms %.3f' % r['avg_launch_ms'])"
  grep "^\[front\]" gpurun_out/r03_b11.err | tail -1
done
