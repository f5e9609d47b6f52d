# kernel stats of one pass of 100k tracks over the solved 10 m field (rocprofv3 --kernel-trace --stats)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
rm -rf /tmp/psol
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d /tmp/psol -o p --output-format csv -- python3 bench.py --potential solve --steps 1 --warmup 0 --cpu-seconds 0 --solved-tracks 0 --no-chain-probe > /tmp/psol.log 2>&1 || { tail -5 /tmp/psol.log; exit 1; }
mkdir -p gpurun_out/prof_solved
cp $(find /tmp/psol -name '*kernel_stats.csv' | head -1) gpurun_out/prof_solved/kernel_stats.csv
python3 tools/summarize_profile.py /tmp/psol "r02 solved field: rocprofv3 --kernel-trace --stats -- python3 bench.py --potential solve --steps 1 --warmup 0 (100k tracks, one pass; MI355X)" > gpurun_out/prof_solved/kernel_stats.md
head -24 gpurun_out/prof_solved/kernel_stats.md
