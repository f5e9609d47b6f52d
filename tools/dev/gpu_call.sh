#!/bin/bash
# One GPU call, parametrised (replaces the one-shot r03_call*.sh wrappers):
#   TAG=r04_x TESTS=all|none|"<pytest -k expression>"|"<test files>" SOAK=<seconds> BENCH="<bench.py args>"|none \
#   EXTRA="<command run last>" bash tools/dev/gpu_call.sh
# Every step writes under gpurun_out/$TAG/; a failing step ends the call (no GPU step after a failure).
cd "${GRAFT_REPO_ROOT:-$PWD}" || exit 1
export TMPDIR=/tmp
TAG=${TAG:-r04_call}; TESTS=${TESTS:-all}; SOAK=${SOAK:-0}; BENCH=${BENCH:-none}
OUT=gpurun_out/$TAG; mkdir -p "$OUT"
if [ "$TESTS" != none ]; then
    if [ "$TESTS" = all ]; then sel=(tests); elif [[ "$TESTS" == tests/* ]]; then sel=($TESTS); else sel=(tests -k "$TESTS"); fi
    timeout -k 10 ${TEST_TIMEOUT:-1000} python -m pytest "${sel[@]}" ${PYTEST_X--x} -q -m gpu --durations=15 > "$OUT/tests.log" 2>&1
    rc=$?; echo "tests rc $rc"; tail -${TEST_TAIL:-8} "$OUT/tests.log"; [ $rc -eq 0 ] || exit 1
fi
if [ "$SOAK" != 0 ]; then
    timeout -k 10 $((SOAK + 120)) python tests/dev/soak_tracks.py "$SOAK" > "$OUT/soak.log" 2>&1
    rc=$?; echo "soak rc $rc"; tail -3 "$OUT/soak.log"; [ $rc -eq 0 ] || exit 1
fi
if [ "$BENCH" != none ]; then
    timeout -k 10 ${BENCH_TIMEOUT:-500} python bench.py $BENCH > "$OUT/bench.json" 2> "$OUT/bench.err"
    rc=$?; echo "bench rc $rc"; [ $rc -eq 0 ] || { tail -8 "$OUT/bench.err"; exit 1; }
    python - "$OUT/bench.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k in ('value', 'unit', 'ms_per_step', 'steps_per_s', 'raster_mcells_per_s', 'steps_per_track_mean', 'share_at_max_moves'):
    print(k, d.get(k))
print('roofline', {k: v for k, v in d['roofline'].items() if k != 'dependent_chain'})
print('solver', d.get('solver'))
c = d.get('cpu_baseline', {})
print('cpu', {k: c.get(k) for k in ('value', 'unit', 'cores', 'sample_lengths_equal_gpu', 'sample_histogram_equal_gpu')})
PY
fi
if [ -n "$EXTRA" ]; then
    bash -c "$EXTRA" > "$OUT/extra.log" 2>&1; rc=$?; echo "extra rc $rc"; tail -${EXTRA_TAIL:-30} "$OUT/extra.log"; [ $rc -eq 0 ] || exit 1
fi
exit 0
