#!/bin/bash
# last GPU call of round 4: the final source -- sampler bench, smoke, the whole -m gpu suite
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $R; cd $GRAFT_REPO_ROOT
timeout -k 10 200 python tools/sampler_bench.py 2>/dev/null > $R/r04_sampler_bench.jsonl; grep '"rows": 1,' $R/r04_sampler_bench.jsonl | cut -c1-130
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $R/r04_smoke.log
timeout -k 10 900 python -m pytest tests -m gpu -q --timeout 600 2>&1 | tail -3 | tee $R/r04_pytest_gpu_tail.log
