cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
python tools/measure_simulator_c2.py 2>&1 | grep -v amdgpu | grep -i "potential," > gpurun_out/r04_sim_c2.txt; tail -4 gpurun_out/r04_sim_c2.txt | cut -c1-250
python tools/measure_c5_share.py 1 100000 2>&1 | grep -v amdgpu | tail -4 > gpurun_out/r04_c4_share.txt; cat gpurun_out/r04_c4_share.txt | cut -c1-300
for n in 125000 250000 1000000; do
  python bench.py --tracks $n --steps 1 --warmup 0 --cpu-seconds 0 --stand-in-steps 0 --no-chain-probe 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print(f\"{d['config']['tracks_total']} tracks: {d['ms_per_step']:.1f} ms per pass, {d['value']:.3e} tracks/s, {d['steps_per_s']:.3e} steps/s ({d['launches_per_step']['pair_table']} pair-table launches; in the kernel {r['steps_per_s_in_kernel']:.3e}, VALU issue frac {r.get('valu_issue_frac')}, lanes {r.get('live_lanes_per_wave')})\")"
done | tee gpurun_out/r04_tracks_sweep.txt
