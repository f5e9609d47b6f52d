# rocprofv3 summaries of one solved-field bench pass: kernel stats + PMC passes (SQ, TCC hit/miss, FETCH, WRITE)
# usage: prof_solved_pass.sh TAG ["extra bench.py args, e.g. --tracks 250000"] ["what the pass is, for the titles"]
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
TAG=${1:-r03_solved}
CMD="python3 bench.py --steps 1 --warmup 0 --cpu-seconds 0 --stand-in-steps 0 --no-chain-probe --full-chip-tracks 0 $2"
WHAT=${3:-100k tracks through the K5 field, one pass incl. the solve; MI355X}
rm -rf /tmp/psol; timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/psol -o p --output-format csv -- $CMD > /tmp/psol.log 2>&1 || { tail -5 /tmp/psol.log; exit 1; }
mkdir -p gpurun_out/$TAG
cp $(find /tmp/psol -name '*kernel_stats.csv' | head -1) gpurun_out/$TAG/kernel_stats.csv
python3 tools/summarize_profile.py /tmp/psol "$TAG: rocprofv3 --kernel-trace --stats -- $CMD ($WHAT)" > gpurun_out/$TAG/kernel_stats.md
head -16 gpurun_out/$TAG/kernel_stats.md
i=0
for pass in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "WRITE_SIZE" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_TA_BUSY_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1)); rm -rf /tmp/pm$i
  timeout -k 10 300 rocprofv3 --pmc $pass --kernel-trace -d /tmp/pm$i -o p --output-format csv -- $CMD > /tmp/pm$i.log 2>&1 || { echo "pmc pass $i failed"; tail -3 /tmp/pm$i.log; }
done
python3 tools/summarize_counters.py "$TAG counters (separate --pmc passes over: $CMD)" /tmp/pm1 /tmp/pm2 /tmp/pm5 > gpurun_out/$TAG/counters.md
python3 tools/summarize_pmc.py /tmp/pm3 /tmp/pm4 gpurun_out/$TAG/pmc_traffic.json "$TAG HBM-side traffic" > gpurun_out/$TAG/pmc_traffic.md
grep "k_step_roam\|k_roam_build\|^| kernel" gpurun_out/$TAG/counters.md | cut -c1-700
grep "k_step_roam\|k_roam_build\|^| kernel" gpurun_out/$TAG/pmc_traffic.md
