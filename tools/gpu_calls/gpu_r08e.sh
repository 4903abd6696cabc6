TAG=r08e STAGES="tests" TESTS="tests/test_moe.py" TESTS_TIMEOUT=900 tools/gpu_run.sh
echo "== mixtral b1 head / S8_SINGLE"
for v in "" "EXL2_LEAN_S8_SINGLE=1"; do
  env $v python bench.py --model mixtral-8x7b --recipe 3.5bpw --steps 32 --warmup 4 --headline-only --no-parity-check 2>/dev/null | cut -c1-200
done
TAG=r08e STAGES="ab" AB_NAME=7b_s8single REPS=2 VARIANTS="head s8=EXL2_LEAN_S8_SINGLE=1" tools/gpu_run.sh
TAG=r08e STAGES="ab" AB_NAME=70b_s8single REPS=1 AB_STEPS=48 AB_FLAGS="--model llama2-70b --recipe 2.5bpw --cache q4" VARIANTS="head s8=EXL2_LEAN_S8_SINGLE=1" tools/gpu_run.sh
(cd /tmp && EXL2_LEAN_S8_SINGLE=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/prof_mx -o r08e -- python /root/repo/bench.py --model mixtral-8x7b --recipe 3.5bpw --steps 24 --warmup 4 --windows 1 --headline-only --no-parity-check > /root/repo/gpurun_out/r08e_mx_rocprof.log 2>&1)
f=$(find gpurun_out/prof_mx -name "r08e_kernel_stats.csv" | head -1); cp $f gpurun_out/r08e_mixtral_b1_s8single_kernel_stats.csv; head -8 $f | cut -c1-200; rm -rf gpurun_out/prof_mx
