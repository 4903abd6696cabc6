#!/bin/bash
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $R; cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_sampling.py -m gpu -q 2>&1 | tail -2
for q in 1 0 1 0; do echo "EXL2_SAMPLE_QUICK=$q"; EXL2_SAMPLE_QUICK=$q timeout -k 10 200 python tools/sampler_bench.py 2>/dev/null | grep '"vocab": 32000, "rows": 1,' | cut -c1-140; done | tee $R/r04_sampler_quick_ab.txt
timeout -k 10 200 python tools/sampler_bench.py 2>/dev/null > $R/r04_sampler_bench.jsonl
