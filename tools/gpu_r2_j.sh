#!/bin/bash
# Round 2: the vectorised sampler kernel -- its parity tests on the GPU and its timing.
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
echo "== sampler tests"; timeout -k 5 100 python -m pytest tests/test_sampling.py tests/test_abi.py -m gpu -q --timeout 90 > $R/r02j_pytest_sampling.log 2>&1; echo "rc=$?"; tail -3 $R/r02j_pytest_sampling.log
echo "== sampler bench"; timeout -k 5 60 python tools/sampler_bench.py > $R/r02_sampler_bench_v2.jsonl 2> $R/r02_sampler_bench.err; echo "rc=$?"; cut -c1-200 $R/r02_sampler_bench_v2.jsonl
