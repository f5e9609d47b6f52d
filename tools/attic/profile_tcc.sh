#!/bin/bash
# L2 requests / hits / misses and VMEM instruction counts of the bench's kernels (two --pmc passes)
set -e
ROOT="${GRAFT_REPO_ROOT:-$PWD}"
OUT="$ROOT/gpurun_out/prof_${1:-tcc}"
ARGS="--steps 3 --warmup 1 --cpu-seconds 0 --solved-tracks 0 --no-chain-probe ${BENCH_ARGS:-}"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d "$OUT/sq" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/sq.log" 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum --kernel-trace --output-format csv -d "$OUT/tcc" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/tcc.log" 2>&1 || true
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum --kernel-trace --output-format csv -d "$OUT/tcp" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/tcp.log" 2>&1 || true
cd "$ROOT"
python3 tools/summarize_counters.py "${1:-tcc}" "$OUT/sq" "$OUT/tcc" "$OUT/tcp" | grep -v "k_plan\|k_tracks_init\|k_prior\|k_updraft\|k_transition" > "$OUT/counters.md"
rm -rf "$OUT/sq" "$OUT/tcc" "$OUT/tcp"
cat "$OUT/counters.md"
