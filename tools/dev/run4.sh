set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 400 python tests/dev/soak_tracks.py 150 > gpurun_out/soak4.log 2>&1 || { tail -5 gpurun_out/soak4.log; exit 1; }
tail -1 gpurun_out/soak4.log
timeout -k 10 600 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
python -c "
import json; d=json.loads(open('gpurun_out/bench_default.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['phase_ms_per_step']); print(d['roofline']); print(d['solved_potential']); print(d['cpu_baseline'])"
