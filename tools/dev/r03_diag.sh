cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout -k 10 200 python tools/dev/r03_diag.py 100000 400000 > gpurun_out/r03_diag.txt 2>&1; echo rc $?; tail -4 gpurun_out/r03_diag.txt
rocprofv3 -L > gpurun_out/r03_counters_list.txt 2>&1; grep -c . gpurun_out/r03_counters_list.txt
for pass in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TA_BUSY_avr TA_TA_BUSY_sum"; do
  tag=$(echo $pass | cut -d' ' -f1); rm -rf /tmp/pm_$tag
  timeout -k 10 250 rocprofv3 --pmc $pass --kernel-trace -d /tmp/pm_$tag -o p --output-format csv -- python3 tools/dev/r03_diag.py 100000 400000 > /tmp/pm_$tag.log 2>&1 || { echo "pmc $tag failed"; tail -3 /tmp/pm_$tag.log; }
done
python3 tools/summarize_counters.py "r03 roam kernel counters (100k tracks, cap 400k)" /tmp/pm_SQ_WAVE_CYCLES /tmp/pm_TCC_HIT_sum /tmp/pm_TCP_TOTAL_CACHE_ACCESSES_sum > gpurun_out/r03_diag_counters.md 2>&1; grep "k_step_roam\|k_step_thr\|^| kernel" gpurun_out/r03_diag_counters.md | cut -c1-600
