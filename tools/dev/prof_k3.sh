# kernel-only durations of the binning kernel (rocprofv3), product + probe libraries (python -O: the probes' histograms are wrong)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for lib in ${LIBS:-libssrs_hip libssrs_probe_k3_noflush libssrs_probe_k3_noread}; do
  [ -f ssrs_amd/$lib.so ] || continue
  rm -rf /tmp/pk3; export SSRS_ALLOW_PROBE_LIB=1 SSRS_HIP_LIB=$PWD/ssrs_amd/$lib.so
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/pk3 -o p --output-format csv -- python3 -O bench.py --cpu-seconds 0 --solved-tracks 0 --no-chain-probe --steps 3 --warmup 1 > /tmp/pk3.log 2>&1 || { tail -5 /tmp/pk3.log; }
  f=$(find /tmp/pk3 -name '*kernel_stats.csv' | head -1)
  echo "$lib: $(grep k_bin_visits16 $f | sed "s/.*)\",//" )"
done
