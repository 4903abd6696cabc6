#!/bin/bash
# Round 2, last call: the sampler kernel after the parallel digit selection -- parity tests on the GPU, timing, smoke().
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
echo "== sampler tests"; timeout -k 5 60 python -m pytest tests/test_sampling.py -m gpu -q --timeout 50 > $R/r02j_pytest_sampling.log 2>&1; echo "rc=$?"; tail -3 $R/r02j_pytest_sampling.log
echo "== sampler bench"; timeout -k 5 30 python tools/sampler_bench.py > $R/r02_sampler_bench_v3.jsonl 2> $R/r02_sampler_bench.err; echo "rc=$?"; cut -c1-200 $R/r02_sampler_bench_v3.jsonl
echo "== smoke"; timeout -k 5 30 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
