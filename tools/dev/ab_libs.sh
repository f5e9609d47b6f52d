# bench step with each library in LIBS (A/B of probe builds)
cd $GRAFT_REPO_ROOT
for lib in $LIBS; do
  SSRS_ALLOW_PROBE_LIB=1 SSRS_HIP_LIB=$PWD/ssrs_amd/$lib.so timeout -k 10 300 python bench.py --cpu-seconds 0 --solved-tracks 0 --no-chain-probe --steps 10 > gpurun_out/ab.json 2> gpurun_out/ab.err || { tail -3 gpurun_out/ab.err; continue; }
  python -c "
import json; d=json.loads(open('gpurun_out/ab.json').read().strip().splitlines()[-1]); print('$lib', round(d['ms_per_step'],3), round(d['phase_ms_per_step']['stepper_kernels_k2b'],3))"
done
