set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_tracks.py tests/test_gpu_g10.py -x -q -m gpu > gpurun_out/check_solved.log 2>&1 || { tail -30 gpurun_out/check_solved.log; exit 1; }
tail -2 gpurun_out/check_solved.log
timeout -k 10 400 python tests/dev/soak_tracks.py 100 > gpurun_out/check_solved_soak.log 2>&1 || { tail -5 gpurun_out/check_solved_soak.log; exit 1; }
tail -1 gpurun_out/check_solved_soak.log
SSRS_TRACKS_DEBUG=1 timeout -k 10 600 python bench.py --potential solve --steps 1 --warmup 0 --cpu-seconds 0 --solved-tracks 0 --no-chain-probe > gpurun_out/bs3.json 2> gpurun_out/bs3.err
python -c "
import json; d=json.loads(open('gpurun_out/bs3.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['steps_per_s'], d['phase_ms_per_step'], d['roofline']['launches'])"
grep "tracks\] launch" gpurun_out/bs3.err | sed -n "1,14p"
