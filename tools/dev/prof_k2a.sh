# kernel-only durations of the table builder (rocprofv3), product + probe libraries
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for lib in ${LIBS:-libssrs_hip libssrs_probe_k2a_nostore libssrs_probe_k2a_noload}; do
  [ -f ssrs_amd/$lib.so ] || continue
  rm -rf /tmp/pk2a; export SSRS_ALLOW_PROBE_LIB=1 SSRS_HIP_LIB=$PWD/ssrs_amd/$lib.so
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/pk2a -o p --output-format csv -- python3 tools/dev/time_k2a.py > /tmp/pk2a.log 2>&1 || { tail -5 /tmp/pk2a.log; exit 1; }
  f=$(find /tmp/pk2a -name '*kernel_stats.csv' | head -1)
  echo "$lib: $(grep transition_thr $f | sed "s/.*)\",//" )"; head -1 $f
done
