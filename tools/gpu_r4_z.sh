#!/bin/bash
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $R; cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_sampling.py tests/test_dropin_reference.py -m gpu -q 2>&1 | tail -2
timeout -k 10 200 python tools/sampler_bench.py 2>/dev/null > $R/r04_sampler_bench.jsonl; cut -c1-130 $R/r04_sampler_bench.jsonl
