cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
# per-launch durations of the front kernels, with and without the staged rows
for v in lr nolr; do
  if [ $v = nolr ]; then export SSRS_TRACKS_NO_LDS_ROWS=1; fi
  rm -rf /tmp/prof_$v
  rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_$v -- python bench.py --potential ramp --steps 2 --warmup 1 --cpu-seconds 0 --no-chain-probe > /dev/null 2>&1
  f=$(find /tmp/prof_$v -name "*kernel_trace.csv" | head -1)
  python - "$f" $v <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
out = []
for r in rows:
    n = r['Kernel_Name']
    if 'k_step_thr' in n or 'k_step_tracks' in n:
        out.append((n[:60], (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, r.get('LDS_Block_Size', ''), r.get('VGPR_Count', '')))
print(sys.argv[2], 'last pass:')
for o in out[-7:]:
    print('   %-60s %8.1f us  lds %s vgpr %s' % o)
PY
done
exit 0
