#!/bin/bash
# after the sliced-ELL sweeps + the early exit from a stagnating BiCGStab: full GPU suite, both potential soaks, C5 share, C2
cd "${GRAFT_REPO_ROOT:-$PWD}" || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/r04_sell; mkdir -p "$OUT"
timeout -k 10 900 python -m pytest tests -x -q -m gpu --durations=5 > "$OUT/tests_full.log" 2>&1; rc=$?
tail -9 "$OUT/tests_full.log"; [ $rc -eq 0 ] || exit 1
python tests/dev/soak_potential.py 100 777 > "$OUT/soak777.txt" 2>&1; rc=$?; tail -2 "$OUT/soak777.txt"; [ $rc -eq 0 ] || exit 1
python tests/dev/soak_potential.py 100 4242 > "$OUT/soak4242.txt" 2>&1; rc=$?; tail -2 "$OUT/soak4242.txt"; [ $rc -eq 0 ] || exit 1
python tools/measure_c5_share.py 32 10000 > "$OUT/c5_share.txt" 2>&1; rc=$?; grep -v amdgpu "$OUT/c5_share.txt" | tail -6 | cut -c1-400; [ $rc -eq 0 ] || exit 1
python tools/measure_c5_share.py 1 100000 2>&1 | grep -v amdgpu | tail -4 | cut -c1-300 > "$OUT/c4_share.txt"; cat "$OUT/c4_share.txt"
