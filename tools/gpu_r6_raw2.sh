#!/bin/bash
# round 6: the lean kernel with LEAN_RAW2 (tools/build_variant.sh raw2 -DLEAN_RAW2=1): parity (chain + qmatrix tests on the variant), then same-box A/B
L=$GRAFT_REPO_ROOT/exllamav2_amd/libexl2_hip_raw2.so
mkdir -p gpurun_out
EXL2_HIP_LIB=$L timeout -k 10 900 python -m pytest tests/test_chain.py tests/test_qmatrix.py tests/test_model.py -m gpu -q --timeout 600 > gpurun_out/r06g_pytest_raw2.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r06g_pytest_raw2.log
TAG=r06g STAGES="ab" AB_NAME=raw2 REPS=3 VARIANTS="head raw2=EXL2_HIP_LIB=$L" tools/gpu_run.sh
EXL2_HIP_LIB=$L timeout -k 10 300 python bench.py --steps 64 --warmup 8 --headline-only 2>/dev/null | tail -1 | cut -c1-1500 > gpurun_out/r06g_bench_raw2.json; cut -c1-900 gpurun_out/r06g_bench_raw2.json
