#!/bin/bash
# After the 512-lane roaming blocks: full GPU suite, bench at 1 M tracks on one GPU, default bench line.
cd "${GRAFT_REPO_ROOT:-$PWD}" || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/r04_wide; mkdir -p "$OUT"
timeout -k 10 900 python -m pytest tests -x -q -m gpu --durations=8 > "$OUT/tests_full.log" 2>&1; rc=$?
tail -12 "$OUT/tests_full.log"; [ $rc -eq 0 ] || exit 1
timeout -k 10 300 python bench.py --tracks 1000000 --steps 1 --warmup 0 > "$OUT/bench_1m.json" 2> "$OUT/bench_1m.err" || { tail -5 "$OUT/bench_1m.err"; exit 1; }
python -c "
import json; d=json.load(open('$OUT/bench_1m.json')); print('1M:', d['value'], d['unit'], d['ms_per_step'], 'ms', d.get('steps_per_s'))"
timeout -k 10 400 python bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_driver_flags.json" 2> "$OUT/bench.err" || { tail -5 "$OUT/bench.err"; exit 1; }
python -c "
import json; d=json.load(open('$OUT/bench_driver_flags.json')); print('headline:', d['value'], d['unit'], d['ms_per_step'], 'ms'); print({k:v for k,v in d['roofline'].items() if k!='dependent_chain'})"
