cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for v in "" "SSRS_HIP_LIB=$GRAFT_REPO_ROOT/ssrs_amd/libssrs_tmp_CHAIN8.so" "SSRS_HIP_LIB=$GRAFT_REPO_ROOT/ssrs_amd/libssrs_tmp_IND8.so"; do
echo "== $v"
env $v timeout -k 10 300 python bench.py --steps 2 --warmup 1 --cpu-seconds 0 --stand-in-steps 0 --no-chain-probe 2>/dev/null > gpurun_out/r03_bench18.json; python - <<'PY'
import json
d = json.loads(open('gpurun_out/r03_bench18.json').read().strip().splitlines()[-1])
r = d['roofline']
print('ms/pass %.1f' % d['ms_per_step'], 'roam launches', r['launches'], 'avg ms %.3f' % r['avg_launch_ms'], 'in-kernel steps/s %.4e' % r['steps_per_s_in_kernel'], 'wave pairs', d['roam']['wave_pairs'])
PY
done
