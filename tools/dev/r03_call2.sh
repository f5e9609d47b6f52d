# roam table: stepper tests, at-size test, soak, bench
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout -k 10 700 python -m pytest tests/test_gpu_tracks.py tests/test_gpu_roaming_c2.py tests/test_gpu_g10.py -x -q -m gpu > gpurun_out/r03_t2.log 2>&1; rc=$?; echo "tests rc $rc"; tail -15 gpurun_out/r03_t2.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python tests/dev/soak_tracks.py 150 > gpurun_out/r03_soak2.log 2>&1; rc=$?; echo "soak rc $rc"; tail -2 gpurun_out/r03_soak2.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 400 python bench.py --steps 2 --warmup 1 --stand-in-steps 0 --no-chain-probe --cpu-seconds 10 > gpurun_out/r03_bench1.json 2> gpurun_out/r03_bench1.err; echo "bench rc $?"; tail -3 gpurun_out/r03_bench1.err; python -c "
import json; d=json.load(open('gpurun_out/r03_bench1.json'))
for k in ('value','ms_per_step','steps_per_s','steps_per_track_mean','share_at_max_moves','launches_per_step','phase_ms_per_step'): print(k, d.get(k))
print('roofline', {k:v for k,v in d['roofline'].items() if k!='dependent_chain'})
c=d.get('cpu_baseline',{}); print({k:c.get(k) for k in ('sample_lengths_equal_gpu','sample_histogram_equal_gpu','sample_gpu_stats','sample_finished_lengths_equal_timed_pass')})
"
SSRS_TRACKS_DEAL_ROUND_ROBIN=1 timeout -k 10 300 python bench.py --steps 1 --warmup 1 --stand-in-steps 0 --no-chain-probe --cpu-seconds 0 > gpurun_out/r03_bench1_rr.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/r03_bench1_rr.json')); print('round-robin deal:', d['value'], d['ms_per_step'], d['steps_per_s'])"
