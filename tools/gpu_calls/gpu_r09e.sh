TAG=r09e STAGES="tests" TESTS="tests/test_chain.py tests/test_moe.py tests/test_qmatrix.py" TESTS_TIMEOUT=900 tools/gpu_run.sh
TAG=r09e STAGES="ab" AB_NAME=70b_duo REPS=2 AB_STEPS=48 AB_FLAGS="--model llama2-70b --recipe 2.5bpw --cache q4" VARIANTS="head noduo=EXL2_HIP_LIB=/root/repo/exllamav2_amd/libexl2_hip_noduo.so" tools/gpu_run.sh
TAG=r09e STAGES="ab" AB_NAME=mixtral_duo REPS=2 AB_STEPS=48 AB_FLAGS="--model mixtral-8x7b --recipe 3.5bpw" VARIANTS="head noduo=EXL2_HIP_LIB=/root/repo/exllamav2_amd/libexl2_hip_noduo.so" tools/gpu_run.sh
TAG=r09e STAGES="ab" AB_NAME=7b_duo REPS=2 VARIANTS="head noduo=EXL2_HIP_LIB=/root/repo/exllamav2_amd/libexl2_hip_noduo.so" tools/gpu_run.sh
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/prof_70b -o r09e -- python /root/repo/bench.py --model llama2-70b --recipe 2.5bpw --cache q4 --steps 24 --warmup 4 --windows 1 --headline-only --no-parity-check > /root/repo/gpurun_out/r09e_70b_rocprof.log 2>&1)
f=$(find gpurun_out/prof_70b -name "r09e_kernel_stats.csv" | head -1); cp $f gpurun_out/r09e_70b_kernel_stats.csv; head -6 $f | cut -c1-200; rm -rf gpurun_out/prof_70b
