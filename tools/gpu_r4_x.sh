#!/bin/bash
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $R; cd $GRAFT_REPO_ROOT
for o in 0 32 0 32; do echo -n "OPT=$o : "; EXL2_SAMPLE_OPT=$o timeout -k 10 200 python tools/sampler_bench.py 2>/dev/null | grep '"vocab": 32000, "rows": 1,' | python -c "
import sys, json
print(' | '.join(f\"{json.loads(l)['setting'][:14]} {json.loads(l)['sample_rows_us']}\" for l in sys.stdin))"; done | tee $R/r04_sampler_exp_ab.txt
EXL2_SAMPLE_OPT=32 timeout -k 10 300 python -m pytest tests/test_sampling.py -m gpu -q 2>&1 | tail -2
