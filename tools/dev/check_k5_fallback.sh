#!/bin/bash
# The K5 fall-back's checks in one GPU call: the soak case that needs it, the configurations that must not take it.
cd "${GRAFT_REPO_ROOT:-$PWD}" || exit 1
OUT=gpurun_out/r04_fallback; mkdir -p "$OUT"
SSRS_PROGRESS=1 python tests/dev/soak_potential_one.py 342679122 2>&1 | grep -v "PCG it\|BiCGStab it [0-9]* |r|\|sum r\|after PCG" > "$OUT/case_342679122.txt" || exit 1
tail -25 "$OUT/case_342679122.txt"
python tools/dev/probe_k5.py 5000x6000 "default" > "$OUT/c2.txt" 2>&1 || exit 1
tail -4 "$OUT/c2.txt"
python tools/dev/probe_k5_snapshot.py 25 > "$OUT/snap25.txt" 2>&1 || exit 1
tail -4 "$OUT/snap25.txt"
python tests/dev/soak_potential.py ${1:-150} 777 > "$OUT/soak777.txt" 2>&1; rc=$?
tail -5 "$OUT/soak777.txt"; [ $rc -eq 0 ] || exit 1
python tests/dev/soak_potential.py ${1:-150} 4242 > "$OUT/soak4242.txt" 2>&1; rc=$?
tail -3 "$OUT/soak4242.txt"; exit $rc
