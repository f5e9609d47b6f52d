set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_tracks.py -x -q -m gpu -k "thr" > gpurun_out/t1.log 2>&1 || { tail -30 gpurun_out/t1.log; exit 1; }
tail -3 gpurun_out/t1.log
timeout -k 10 400 python tests/dev/soak_tracks.py 150 > gpurun_out/soak1.log 2>&1 || { tail -5 gpurun_out/soak1.log; exit 1; }
tail -2 gpurun_out/soak1.log
timeout -k 10 300 python bench.py --cpu-seconds 0 --solved-tracks 0 --no-chain-probe > gpurun_out/b1.json 2> gpurun_out/b1.err
python -c "
import json; d=json.loads(open('gpurun_out/b1.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('phases_ms') or d.get('phases'))"
