set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_tracks.py tests/test_gpu_integration_stub.py tests/test_gpu_g10.py -x -q -m gpu > gpurun_out/check_ramp.log 2>&1 || { tail -30 gpurun_out/check_ramp.log; exit 1; }
tail -3 gpurun_out/check_ramp.log
timeout -k 10 400 python tests/dev/soak_tracks.py 120 > gpurun_out/check_ramp_soak.log 2>&1 || { tail -5 gpurun_out/check_ramp_soak.log; exit 1; }
tail -1 gpurun_out/check_ramp_soak.log
timeout -k 10 300 python bench.py --cpu-seconds 0 --solved-tracks 0 --no-chain-probe --steps 10 > gpurun_out/b2.json 2> gpurun_out/b2.err
python -c "
import json; d=json.loads(open('gpurun_out/b2.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['phase_ms_per_step'], d['roofline']['launches'], d['roofline']['avg_launch_ms'])"
