#!/bin/bash
# 512- / 1024-lane roaming blocks against the 256-lane ones: identical integers (digests), steps/s by batch size.
# usage: check_roam_wide.sh [CAP]   (CAP=0: max_moves as configured, 7.5e6)
cd "${GRAFT_REPO_ROOT:-$PWD}" || exit 1
OUT=gpurun_out/r04_wide; mkdir -p "$OUT"; F="$OUT/fill_cap${1:-0}.txt"; : > "$F"
export CAP=${1:-0}
for w in 1 2 4; do
  echo "# SSRS_TRACKS_ROAM_WIDTH=$w" >> "$F"
  SSRS_TRACKS_ROAM_WIDTH=$w timeout -k 10 500 python tools/dev/roam_fill.py ${SIZES:-140000 200000 280000 400000 500000} >> "$F" 2>&1 || { tail -20 "$F"; exit 1; }
done
echo "# default" >> "$F"
timeout -k 10 500 python tools/dev/roam_fill.py 100000 ${SIZES:-140000 200000 280000 400000 500000} >> "$F" 2>&1 || { tail -20 "$F"; exit 1; }
grep -v "amdgpu.ids\|^# library" "$F"
