#!/bin/bash
# round 2, call A: every -m gpu test (formerly gated hip variants included), then the headline bench with the pre-timing oracle check
mkdir -p gpurun_out; export TMPDIR=/tmp; R=gpurun_out
echo "== pytest gpu"; timeout 900 python -m pytest tests -m gpu -q --timeout 600 -x > $R/r2a_pytest_gpu.log 2>&1; echo "rc=$?"; tail -15 $R/r2a_pytest_gpu.log
echo "== bench"; timeout 600 python bench.py --no-prefill --no-cpu-baseline > $R/r2a_bench.json 2> $R/r2a_bench.err; echo "rc=$?"; cut -c1-2000 $R/r2a_bench.json; tail -5 $R/r2a_bench.err
