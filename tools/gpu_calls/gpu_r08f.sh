TAG=r08f STAGES="tests" TESTS="tests/test_moe.py tests/test_chain.py tests/test_model.py" TESTS_TIMEOUT=900 tools/gpu_run.sh
echo "== mixtral b1: sum route (with parity) / two down launches + combine"
python bench.py --model mixtral-8x7b --recipe 3.5bpw --steps 32 --warmup 4 --headline-only > gpurun_out/r08f_bench_mixtral_b1.json 2> gpurun_out/r08f_mixtral.err; cut -c1-1300 gpurun_out/r08f_bench_mixtral_b1.json; tail -2 gpurun_out/r08f_mixtral.err
EXL2_MOE_NO_SUM=1 python bench.py --model mixtral-8x7b --recipe 3.5bpw --steps 32 --warmup 4 --headline-only --no-parity-check 2>/dev/null | cut -c1-200
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/prof_mx -o r08f -- python /root/repo/bench.py --model mixtral-8x7b --recipe 3.5bpw --steps 24 --warmup 4 --windows 1 --headline-only --no-parity-check > /root/repo/gpurun_out/r08f_mx_rocprof.log 2>&1)
f=$(find gpurun_out/prof_mx -name "r08f_kernel_stats.csv" | head -1); cp $f gpurun_out/r08f_mixtral_b1_kernel_stats.csv; head -8 $f | cut -c1-200; rm -rf gpurun_out/prof_mx
