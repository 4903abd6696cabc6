set -x
TAG=r08a STAGES="tests" TESTS="tests/test_chain.py" TESTS_TIMEOUT=900 tools/gpu_run.sh
TAG=r08a STAGES="ab" AB_NAME=7b REPS=3 VARIANTS="head old=EXL2_HIP_LIB=/root/repo/exllamav2_amd/libexl2_hip_oldlean.so" tools/gpu_run.sh
TAG=r08a STAGES="ab" AB_NAME=70b REPS=2 AB_STEPS=48 AB_FLAGS="--model llama2-70b --recipe 2.5bpw --cache q4" VARIANTS="head old=EXL2_HIP_LIB=/root/repo/exllamav2_amd/libexl2_hip_oldlean.so" tools/gpu_run.sh
cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/prof_70b -o r08a -- python /root/repo/bench.py --model llama2-70b --recipe 2.5bpw --cache q4 --steps 24 --warmup 4 --windows 1 --headline-only --no-parity-check > /root/repo/gpurun_out/r08a_70b_rocprof.log 2>&1
f=$(find /root/repo/gpurun_out/prof_70b -name "r08a_kernel_stats.csv" | head -1); cp $f /root/repo/gpurun_out/r08a_70b_kernel_stats.csv; head -8 $f | cut -c1-200; rm -rf /root/repo/gpurun_out/prof_70b
