#!/bin/bash
# Round 3: how much of a decode step is the bit-width / group-size mix?  Same model, all linears 4-bit g128 (4.0bpw_plain).
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT/gpurun_out; cd $GRAFT_REPO_ROOT
B="python bench.py --steps 64 --warmup 8 --no-cpu-baseline --no-prefill --no-ctx-window --no-parity-check"
for r in 4.0bpw 4.0bpw_plain; do timeout -k 10 300 $B --recipe $r 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); rf=d.get('roofline',{}); print('$r', d['value'], 'tok/s', rf.get('avg_launch_us'), 'us/launch frac', rf.get('frac'), 'bytes/launch', rf.get('algorithmic_bytes_per_launch'))"; done | tee $R/r03_recipes.txt
