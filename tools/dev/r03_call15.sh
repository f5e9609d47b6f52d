cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_tracks.py tests/test_gpu_roaming_c2.py -x -q -m gpu > gpurun_out/r03_t15.log 2>&1; rc=$?; echo "tests rc $rc"; tail -3 gpurun_out/r03_t15.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python bench.py --steps 2 --warmup 1 --cpu-seconds 0 --stand-in-steps 5 --no-chain-probe 2>/dev/null > gpurun_out/r03_bench15.json; python - <<'PY'
import json
d = json.loads(open('gpurun_out/r03_bench15.json').read().strip().splitlines()[-1])
r = d['roofline']
print('tracks/s %.4e ms/pass %.1f steps/s %.4e' % (d['value'], d['ms_per_step'], d['steps_per_s']), 'roam launches', r['launches'], 'avg ms %.3f' % r['avg_launch_ms'], 'in-kernel steps/s %.4e' % r['steps_per_s_in_kernel'], 'throughput_frac %.3f' % r['throughput_frac'])
print('stand_in', {k: d['stand_in'][k] for k in ('value', 'ms_per_step') if k in d['stand_in']}, d['stand_in'].get('phase_ms_per_step'))
PY
