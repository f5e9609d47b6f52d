set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_raster.py tests/test_gpu_full_chain.py tests/test_gpu_configs.py -x -q -m gpu > gpurun_out/tk1.log 2>&1 || { tail -30 gpurun_out/tk1.log; exit 1; }
tail -2 gpurun_out/tk1.log
timeout -k 10 300 python bench.py --cpu-seconds 0 --solved-tracks 0 --no-chain-probe --steps 10 > gpurun_out/b1.json 2> gpurun_out/b1.err
python -c "
import json; d=json.loads(open('gpurun_out/b1.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['phase_ms_per_step'], d['raster_gbps'])"
