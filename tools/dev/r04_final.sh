#!/bin/bash
# the round's closing run: full GPU suite, soaks, bench (default flags and the driver's), profiles of one pass
cd "${GRAFT_REPO_ROOT:-$PWD}" || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/r04_final; mkdir -p "$OUT"
timeout -k 10 900 python -m pytest tests -x -q -m gpu --durations=5 > "$OUT/tests_full.log" 2>&1; rc=$?
tail -9 "$OUT/tests_full.log"; [ $rc -eq 0 ] || exit 1
timeout -k 10 400 python tests/dev/soak_tracks.py ${1:-200} > "$OUT/soak_tracks.log" 2>&1; rc=$?; tail -1 "$OUT/soak_tracks.log"; [ $rc -eq 0 ] || exit 1
python tests/dev/soak_potential.py ${2:-100} 777 > "$OUT/soak_pot_777.txt" 2>&1; rc=$?; tail -1 "$OUT/soak_pot_777.txt"; [ $rc -eq 0 ] || exit 1
python tests/dev/soak_potential.py ${2:-100} 4242 > "$OUT/soak_pot_4242.txt" 2>&1; rc=$?; tail -1 "$OUT/soak_pot_4242.txt"; [ $rc -eq 0 ] || exit 1
timeout -k 10 400 python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err" || { tail -5 "$OUT/bench_default.err"; exit 1; }
timeout -k 10 400 python bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_driver_flags.json" 2> "$OUT/bench_driver.err" || { tail -5 "$OUT/bench_driver.err"; exit 1; }
python - "$OUT/bench_driver_flags.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print('headline', d['value'], d['unit'], d['ms_per_step'], 'ms; steps/s', d.get('steps_per_s'), 'raster', d.get('raster_mcells_per_s'))
r = d['roofline']; print('roofline', {k: r[k] for k in ('launches', 'avg_launch_ms', 'frac', 'throughput_frac', 'valu_issue_frac', 'gather_path_frac', 'traffic')})
print('solver', d.get('solver')); print('full_chip', {k: v for k, v in d.get('full_chip', {}).items() if k != 'what'})
c = d['cpu_baseline']; print('cpu', {k: c.get(k) for k in ('value', 'unit', 'cores', 'kind', 'sample_lengths_equal_gpu', 'sample_histogram_equal_gpu')})
PY
bash tools/dev/prof_solved_pass.sh r04_final > "$OUT/prof.log" 2>&1; tail -12 "$OUT/prof.log" | cut -c1-400
