TAG=r11b STAGES="tests" TESTS="tests/test_chain.py tests/test_moe.py" TESTS_TIMEOUT=900 tools/gpu_run.sh
TAG=r11b STAGES="ab" AB_NAME=70b_deep REPS=2 AB_STEPS=48 AB_FLAGS="--model llama2-70b --recipe 2.5bpw --cache q4" VARIANTS="head d1=EXL2_HIP_LIB=/root/repo/exllamav2_amd/libexl2_hip_depth1.so" tools/gpu_run.sh
TAG=r11b STAGES="ab" AB_NAME=mixtral_deep REPS=2 AB_STEPS=48 AB_FLAGS="--model mixtral-8x7b --recipe 3.5bpw" VARIANTS="head d1=EXL2_HIP_LIB=/root/repo/exllamav2_amd/libexl2_hip_depth1.so" tools/gpu_run.sh
