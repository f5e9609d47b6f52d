# the round's final measurement set: full -m gpu suite, default bench line, the driver's flags, the solved-field
# profile (kernel stats + PMC passes), a two-rank rehearsal of bench.py over gloo on the one GPU
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
set -o pipefail
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r03_final_tests.log 2>&1; rc=$?; tail -2 gpurun_out/r03_final_tests.log; [ $rc -eq 0 ] || exit 1
timeout -k 10 400 python bench.py > gpurun_out/r03_bench_final.json 2> gpurun_out/r03_bench_final.err; rc=$?; echo "default bench rc $rc"; [ $rc -eq 0 ] || { tail -5 gpurun_out/r03_bench_final.err; exit 1; }
timeout -k 10 400 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03_bench_final_driver_flags.json 2> gpurun_out/r03_bench_final_driver_flags.err; rc=$?; echo "driver-flag bench rc $rc"; [ $rc -eq 0 ] || { tail -5 gpurun_out/r03_bench_final_driver_flags.err; exit 1; }
bash tools/dev/prof_solved_pass.sh r03_solved_final > gpurun_out/r03_prof_final.log 2>&1; rc=$?; echo "profile rc $rc"; tail -6 gpurun_out/r03_prof_final.log | cut -c1-400; [ $rc -eq 0 ] || exit 1
SSRS_BENCH_BACKEND=gloo timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 2 --warmup 1 --cpu-seconds 0 --stand-in-steps 2 --no-chain-probe > gpurun_out/r03_bench_two_ranks_gloo.json 2> gpurun_out/r03_bench_two_ranks_gloo.err; rc=$?; echo "two-rank rehearsal rc $rc"; tail -c 1500 gpurun_out/r03_bench_two_ranks_gloo.json | cut -c1-1500
exit 0
