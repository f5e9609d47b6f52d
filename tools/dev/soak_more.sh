#!/bin/bash
# extra soaks under other master seeds: usage soak_more.sh TRACK_SECONDS POTENTIAL_SECONDS
cd "${GRAFT_REPO_ROOT:-$PWD}" || exit 1
OUT=gpurun_out/r04_soak_more; mkdir -p "$OUT"
timeout -k 10 $(( $1 + 120 )) python tests/dev/soak_tracks.py "$1" 7 > "$OUT/tracks_seed7.log" 2>&1; rc=$?; tail -2 "$OUT/tracks_seed7.log"; [ $rc -eq 0 ] || exit 1
SSRS_TRACKS_ROAM_WIDTH=2 timeout -k 10 $(( $1 / 2 + 120 )) python tests/dev/soak_tracks.py $(( $1 / 2 )) 99 > "$OUT/tracks_w2_seed99.log" 2>&1; rc=$?; tail -2 "$OUT/tracks_w2_seed99.log"; [ $rc -eq 0 ] || exit 1
python tests/dev/soak_potential.py "$2" 31337 > "$OUT/pot_seed31337.txt" 2>&1; rc=$?; tail -3 "$OUT/pot_seed31337.txt"; [ $rc -eq 0 ] || exit 1
python tests/dev/soak_potential.py "$2" 5 > "$OUT/pot_seed5.txt" 2>&1; rc=$?; tail -3 "$OUT/pot_seed5.txt"; exit $rc
