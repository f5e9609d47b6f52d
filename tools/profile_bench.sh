#!/bin/bash
# rocprofv3 passes over the default bench (run on the GPU box from the repo root):
#   kernel stats, SQ wave-time split, L2 hit/miss, HBM-side fetch / write sizes (separate --pmc passes)
set -e
ROOT="${GRAFT_REPO_ROOT:-$PWD}"
OUT="$ROOT/gpurun_out/prof_${1:-r02}"
ARGS="--steps 3 --warmup 1 --cpu-seconds 0 --solved-tracks 0 --no-chain-probe ${BENCH_ARGS:-}"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/stats.log" 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --kernel-trace --output-format csv -d "$OUT/sq" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/sq.log" 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_INSTS_FLAT --kernel-trace --output-format csv -d "$OUT/sq2" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/sq2.log" 2>&1 || true
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d "$OUT/tcc" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/tcc.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/fetch" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/write" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/write.log" 2>&1
cd "$ROOT"
python3 tools/summarize_profile.py "$OUT/stats" "${1:-r02}: rocprofv3 --kernel-trace --stats -- python3 bench.py $ARGS (MI355X)" > "$OUT/kernel_stats.md"
python3 tools/summarize_counters.py "${1:-r02}: SQ / TCC counters (separate --pmc passes), same command" "$OUT/sq" "$OUT/sq2" "$OUT/tcc" > "$OUT/counters.md"
python3 tools/summarize_pmc.py "$OUT/fetch" "$OUT/write" "$OUT/pmc_traffic.json" "${1:-r02} HBM-side traffic" > "$OUT/pmc_traffic.md"
cp "$(ls $OUT/stats/*/*_kernel_stats.csv | head -1)" "$OUT/kernel_stats.csv"
rm -rf "$OUT/stats" "$OUT/sq" "$OUT/sq2" "$OUT/tcc" "$OUT/fetch" "$OUT/write"
cat "$OUT/kernel_stats.md" "$OUT/counters.md" "$OUT/pmc_traffic.md"
