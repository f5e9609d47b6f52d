#!/bin/bash
# The track soak against the C oracle with the roaming blocks' width forced (2 = 512 lanes, 4 = 1024) and by the
# shipped policy (0): usage soak_widths.sh SECONDS_EACH "WIDTHS" [SEED]
cd "${GRAFT_REPO_ROOT:-$PWD}" || exit 1
OUT=gpurun_out/r04_wide; mkdir -p "$OUT"
for w in ${2:-0 2 4}; do
  if [ $w = 0 ]; then unset SSRS_TRACKS_ROAM_WIDTH; else export SSRS_TRACKS_ROAM_WIDTH=$w; fi
  timeout -k 10 $(( $1 + 120 )) python tests/dev/soak_tracks.py "$1" ${3:-} > "$OUT/soak_w$w.log" 2>&1; rc=$?
  echo "width $w rc $rc"; tail -2 "$OUT/soak_w$w.log"; [ $rc -eq 0 ] || exit 1
done
