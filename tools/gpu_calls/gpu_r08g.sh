TAG=r08g STAGES="tests" TESTS="tests/test_chain.py" TESTS_TIMEOUT=900 tools/gpu_run.sh
TAG=r08g STAGES="ab" AB_NAME=70b_rows1 REPS=2 AB_STEPS=48 AB_FLAGS="--model llama2-70b --recipe 2.5bpw --cache q4" VARIANTS="head plain=EXL2_LEAN_ROWS1=0" tools/gpu_run.sh
TAG=r08g STAGES="ab" AB_NAME=7b_rows1 REPS=2 VARIANTS="head plain=EXL2_LEAN_ROWS1=0" tools/gpu_run.sh
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/prof_70b -o r08g -- python /root/repo/bench.py --model llama2-70b --recipe 2.5bpw --cache q4 --steps 24 --warmup 4 --windows 1 --headline-only --no-parity-check > /root/repo/gpurun_out/r08g_70b_rocprof.log 2>&1)
f=$(find gpurun_out/prof_70b -name "r08g_kernel_stats.csv" | head -1); cp $f gpurun_out/r08g_70b_kernel_stats.csv; head -7 $f | cut -c1-200; rm -rf gpurun_out/prof_70b
