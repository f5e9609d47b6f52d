cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
rm -rf /tmp/tr; timeout -k 10 300 rocprofv3 --kernel-trace -d /tmp/tr -o t --output-format csv -- python3 bench.py --cpu-seconds 0 --solved-tracks 0 --no-chain-probe --steps 3 --warmup 1 > /tmp/tr.log 2>&1 || { tail -5 /tmp/tr.log; exit 1; }
f=$(find /tmp/tr -name '*kernel_trace.csv' | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# last bench step: find the last k_updraft_from_dem and print everything after it
idx = max(i for i, r in enumerate(rows) if 'k_updraft_from_dem' in r['Kernel_Name'])
t0 = int(rows[idx]['Start_Timestamp']); prev_end = t0
out = open('gpurun_out/r02_timeline.txt', 'w')
for r in rows[idx:]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    name = r['Kernel_Name'].split('(')[0].replace('void ', '')[:60]
    out.write(f"{(s - t0) / 1e3:9.1f} us  +gap {(s - prev_end) / 1e3:7.1f}  dur {(e - s) / 1e3:8.1f}  {name}\n")
    prev_end = max(prev_end, e)
out.write(f"total {(prev_end - t0) / 1e3:.1f} us\n")
PY
tail -60 gpurun_out/r02_timeline.txt
