#!/bin/bash
# HBM-side traffic and SQ counters of the K5 solve at C2 alone (separate --pmc passes over tools/attic/run_solver_c2.py)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
TAG=${1:-r04_k5_final}; mkdir -p gpurun_out/$TAG
CMD="python3 tools/attic/run_solver_c2.py"
i=0
for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD"; do
  i=$((i+1)); rm -rf /tmp/pk$i
  timeout -k 10 300 rocprofv3 --pmc $pass --kernel-trace -d /tmp/pk$i -o p --output-format csv -- $CMD > /tmp/pk$i.log 2>&1 || { echo "pmc pass $i failed"; tail -3 /tmp/pk$i.log; }
done
python3 tools/summarize_pmc.py /tmp/pk1 /tmp/pk2 gpurun_out/$TAG/pmc_traffic.json "$TAG HBM-side traffic of one K5 solve at C2" > gpurun_out/$TAG/pmc_traffic.md
python3 tools/summarize_counters.py "$TAG counters (separate --pmc passes over: $CMD)" /tmp/pk4 /tmp/pk3 > gpurun_out/$TAG/counters.md
head -24 gpurun_out/$TAG/pmc_traffic.md | cut -c1-160
