cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
# quick parity first (front tests + heading), then the ramp with and without the staged rows
timeout -k 10 400 python -m pytest tests/test_gpu_tracks.py -x -q -m gpu > gpurun_out/r03_t12.log 2>&1; rc=$?; echo "tests rc $rc"; tail -3 gpurun_out/r03_t12.log
[ $rc -eq 0 ] || exit 1
SSRS_TRACKS_LDS_ROWS=1 SSRS_TRACKS_DEBUG_ROAM=1 timeout -k 10 200 python bench.py --potential ramp --steps 1 --warmup 0 --cpu-seconds 0 --no-chain-probe 2> gpurun_out/r03_dbg12.err > /dev/null; grep "^\[front\]" gpurun_out/r03_dbg12.err | tail -3
for v in "" "SSRS_TRACKS_LDS_ROWS=1"; do
  echo "== ramp bench $v"
  env $v timeout -k 10 200 python bench.py --potential ramp --steps 10 --warmup 2 --cpu-seconds 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('tracks/s %.3e' % d['value'], 'ms/step %.3f' % d['ms_per_step'], d['phase_ms_per_step'], 'launches', r['launches'], 'avg launch ms %.3f' % r['avg_launch_ms'])"
done
exit 0
