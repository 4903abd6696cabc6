TAG=r08c STAGES="tests" TESTS="tests/test_moe.py tests/test_chain.py" TESTS_TIMEOUT=900 tools/gpu_run.sh
echo "== mixtral b1 (lean / grouped)"
python bench.py --model mixtral-8x7b --recipe 3.5bpw --steps 32 --warmup 4 --headline-only > gpurun_out/r08c_bench_mixtral_b1.json 2> gpurun_out/r08c_mixtral.err; cut -c1-900 gpurun_out/r08c_bench_mixtral_b1.json; tail -2 gpurun_out/r08c_mixtral.err
EXL2_MOE_NO_LEAN=1 python bench.py --model mixtral-8x7b --recipe 3.5bpw --steps 32 --warmup 4 --headline-only --no-parity-check 2>/dev/null | cut -c1-200
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/prof_mx -o r08c -- python /root/repo/bench.py --model mixtral-8x7b --recipe 3.5bpw --steps 24 --warmup 4 --windows 1 --headline-only --no-parity-check > /root/repo/gpurun_out/r08c_mx_rocprof.log 2>&1)
f=$(find gpurun_out/prof_mx -name "r08c_kernel_stats.csv" | head -1); cp $f gpurun_out/r08c_mixtral_b1_kernel_stats.csv; head -9 $f | cut -c1-200; rm -rf gpurun_out/prof_mx
TAG=r08c STAGES="ab" AB_NAME=7b REPS=2 VARIANTS="head old=EXL2_HIP_LIB=/root/repo/exllamav2_amd/libexl2_hip_oldlean.so" tools/gpu_run.sh
