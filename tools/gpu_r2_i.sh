#!/bin/bash
# Round 2: the full default bench line (cpu_baseline in its child process: unsampled 26 GB pass) + sampler timing.
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
echo "== sampler bench"; timeout -k 5 60 python tools/sampler_bench.py > $R/r02_sampler_bench.jsonl 2> $R/r02_sampler_bench.err; echo "rc=$?"; cut -c1-200 $R/r02_sampler_bench.jsonl; tail -2 $R/r02_sampler_bench.err
echo "== bench (default)"; /usr/bin/time -v timeout -k 5 400 python bench.py > $R/r02_bench.json 2> $R/r02_bench.err; echo "rc=$?"; cut -c1-300 $R/r02_bench.json; python -c "
import json; d=json.loads(open('$R/r02_bench.json').read().strip().splitlines()[-1]); print(json.dumps(d.get('cpu_baseline'))[:1500]); print(d.get('prefill'))"; grep -i "elapsed\|maximum resident" $R/r02_bench.err; free -g | head -2; nproc
