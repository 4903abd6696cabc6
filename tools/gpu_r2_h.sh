#!/bin/bash
# Round 2, last GPU call (5.8 GPU-minutes left): every -m gpu parity test incl. the ones written after the previous call
# (EXL2_RUN_UNVERIFIED=1: device sampler, drop-in sample_basic), then the headline bench (decode + prefill, no CPU baseline).
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
echo "== pytest gpu"; EXL2_RUN_UNVERIFIED=1 timeout -k 5 170 python -m pytest tests -m gpu -q --timeout 120 > $R/r02h_pytest_gpu.log 2>&1; echo "rc=$?"; tail -4 $R/r02h_pytest_gpu.log
echo "== bench"; timeout -k 5 170 python bench.py --no-cpu-baseline > $R/r02h_bench.json 2> $R/r02h_bench.err; echo "rc=$?"; cut -c1-1500 $R/r02h_bench.json; tail -3 $R/r02h_bench.err
