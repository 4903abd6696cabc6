TAG=r10d STAGES="tests" TESTS="tests/test_moe.py" TESTS_TIMEOUT=900 tools/gpu_run.sh
TAG=r10d STAGES="ab" AB_NAME=mixtral_front REPS=3 AB_STEPS=48 AB_FLAGS="--model mixtral-8x7b --recipe 3.5bpw" VARIANTS="head prev=EXL2_HIP_LIB=/root/repo/exllamav2_amd/libexl2_hip_prevfront.so" tools/gpu_run.sh
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/prof_mx -o r10d -- python /root/repo/bench.py --model mixtral-8x7b --recipe 3.5bpw --steps 24 --warmup 4 --windows 1 --headline-only --no-parity-check > /root/repo/gpurun_out/r10d_mx_rocprof.log 2>&1)
f=$(find gpurun_out/prof_mx -name "r10d_kernel_stats.csv" | head -1); cp $f gpurun_out/r10d_mixtral_b1_kernel_stats.csv; head -7 $f | cut -c1-200; rm -rf gpurun_out/prof_mx
