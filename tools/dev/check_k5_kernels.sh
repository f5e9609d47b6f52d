#!/bin/bash
# two-row level-0 passes (+ sliced-ELL sweeps) against the one-row / unfused / CSR variants: solver tests, C2 under both, kernel stats
cd "${GRAFT_REPO_ROOT:-$PWD}" || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/r04_l0two; mkdir -p "$OUT"
timeout -k 10 600 python -m pytest tests -x -q -m gpu -k "potential or g10 or full_chain" > "$OUT/tests.log" 2>&1; rc=$?
tail -4 "$OUT/tests.log"; [ $rc -eq 0 ] || echo "TESTS FAILED (going on: measurements)"
python tools/dev/probe_k5.py 5000x6000 "default;one row;no fuse;no sell" > "$OUT/c2.txt" 2>&1 || { tail "$OUT/c2.txt"; exit 1; }
grep -v amdgpu "$OUT/c2.txt"
python tools/dev/probe_k5_snapshot.py 25 > "$OUT/snap25.txt" 2>&1 || exit 1
head -2 "$OUT/snap25.txt"
rm -rf /tmp/psell; timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/psell -o p --output-format csv -- python3 tools/attic/run_solver_c2.py > /tmp/psell.log 2>&1 || { tail -5 /tmp/psell.log; exit 1; }
python3 tools/summarize_profile.py /tmp/psell "r04 K5 final (two-row fused level 0, sliced-ELL sweeps): rocprofv3 --kernel-trace --stats -- python3 tools/attic/run_solver_c2.py" > "$OUT/kernel_stats.md"
head -14 "$OUT/kernel_stats.md"
