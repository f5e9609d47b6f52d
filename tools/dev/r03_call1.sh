# round 3, first GPU call: the new at-size test, the touched tests, the default bench line, the footprint probe
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout -k 10 500 python -m pytest tests/test_gpu_roaming_c2.py tests/test_gpu_tracks.py::test_thr_table_belongs_to_one_heading tests/test_gpu_integration_stub.py tests/test_gpu_multirank.py -x -q -m gpu > gpurun_out/r03_t1.log 2>&1; echo "tests rc $?"; tail -15 gpurun_out/r03_t1.log
timeout -k 10 400 python bench.py --steps 2 --warmup 1 > gpurun_out/r03_bench0.json 2> gpurun_out/r03_bench0.err; echo "bench rc $?"; tail -3 gpurun_out/r03_bench0.err; python -c "
import json; d=json.load(open('gpurun_out/r03_bench0.json'))
for k in ('value','ms_per_step','steps_per_s','steps_per_track_mean','share_at_max_moves','launches_per_step','phase_ms_per_step','solver'): print(k, d.get(k))
print('roofline', {k:v for k,v in d['roofline'].items() if k!='dependent_chain'})
print('stand_in', {k:v for k,v in d.get('stand_in',{}).items() if k not in ('roofline','potential','what')})
print('cpu', d.get('cpu_baseline'))
"
timeout -k 10 300 python tools/dev/probe_roam_footprint.py > gpurun_out/r03_footprint.txt 2>&1; echo "probe rc $?"; tail -40 gpurun_out/r03_footprint.txt
