cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r03_tall.log 2>&1; rc=$?; echo "tests rc $rc"; tail -6 gpurun_out/r03_tall.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 400 python bench.py --steps 3 --warmup 1 > gpurun_out/r03_bench2.json 2> gpurun_out/r03_bench2.err; echo "bench rc $?"; tail -3 gpurun_out/r03_bench2.err; python -c "
import json; d=json.load(open('gpurun_out/r03_bench2.json'))
for k in ('value','ms_per_step','steps_per_s','steps_per_track_mean','share_at_max_moves','launches_per_step','phase_ms_per_step'): print(k, d.get(k))
print('roofline', {k:v for k,v in d['roofline'].items() if k!='dependent_chain'})
print('stand_in', {k:v for k,v in d.get('stand_in',{}).items() if k not in ('roofline','potential','what')})
c=d.get('cpu_baseline',{}); print({k:c.get(k) for k in ('value','sample_lengths_equal_gpu','sample_histogram_equal_gpu','sample_gpu_stats','sample_finished_lengths_equal_timed_pass','orograph_f32_cells_identical_to_gpu','usable_updraft_max_rel_diff_vs_gpu')})
"
