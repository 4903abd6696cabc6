TAG=r08d STAGES="tests" TESTS="tests/test_moe.py tests/test_model.py tests/test_abi.py" TESTS_TIMEOUT=900 tools/gpu_run.sh
echo "== mixtral b1"
python bench.py --model mixtral-8x7b --recipe 3.5bpw --steps 32 --warmup 4 --headline-only > gpurun_out/r08d_bench_mixtral_b1.json 2> gpurun_out/r08d_mixtral.err; cut -c1-1200 gpurun_out/r08d_bench_mixtral_b1.json; tail -2 gpurun_out/r08d_mixtral.err
echo "== mixtral b1, MoE layers un-chained"
EXL2_CHAIN_MOE=0 python bench.py --model mixtral-8x7b --recipe 3.5bpw --steps 32 --warmup 4 --headline-only --no-parity-check 2>/dev/null | cut -c1-200
echo "== mixtral b4 / b16"
python bench.py --model mixtral-8x7b --recipe 3.5bpw --batch 4 --steps 32 --warmup 4 --headline-only > gpurun_out/r08d_bench_mixtral_b4.json 2>/dev/null; cut -c1-200 gpurun_out/r08d_bench_mixtral_b4.json
python bench.py --model mixtral-8x7b --recipe 3.5bpw --batch 16 --steps 32 --warmup 4 --headline-only > gpurun_out/r08d_bench_mixtral_b16.json 2>/dev/null; cut -c1-200 gpurun_out/r08d_bench_mixtral_b16.json
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/prof_mx -o r08d -- python /root/repo/bench.py --model mixtral-8x7b --recipe 3.5bpw --steps 24 --warmup 4 --windows 1 --headline-only --no-parity-check > /root/repo/gpurun_out/r08d_mx_rocprof.log 2>&1)
f=$(find gpurun_out/prof_mx -name "r08d_kernel_stats.csv" | head -1); cp $f gpurun_out/r08d_mixtral_b1_kernel_stats.csv; head -10 $f | cut -c1-200; rm -rf gpurun_out/prof_mx
