#!/bin/bash
# Round 2: the full default bench line (cpu_baseline in its child process: unsampled 26 GB pass).
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
echo "== bench (default)"; S=$(date +%s); timeout -k 5 400 python bench.py > $R/r02_bench.json 2> $R/r02_bench.err; echo "rc=$? wall=$(( $(date +%s) - S ))s"; cut -c1-300 $R/r02_bench.json; python -c "
import json; d=json.loads(open('$R/r02_bench.json').read().strip().splitlines()[-1]); print(json.dumps(d.get('cpu_baseline'))[:1500]); print(d.get('prefill'))"; tail -3 $R/r02_bench.err
