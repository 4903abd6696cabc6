#!/bin/bash
# Round 3: where a small-batch decode step spends its time (bs = 2, 4, 8): routes + rocprof kernel stats
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT/gpurun_out; cd $GRAFT_REPO_ROOT
for b in ${BATCHES:-2 4 8}; do
  (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/prof_b$b -o b$b -- python $GRAFT_REPO_ROOT/bench.py --batch $b --steps 32 --warmup 4 --no-cpu-baseline --no-prefill --no-parity-check --no-ctx-window > $R/r03_batch$b.log 2>&1)
  tail -1 $R/r03_batch$b.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('bs', $b, d['value'], 'tok/s', d['ms_per_step'], 'ms/step', d.get('extra',{}).get('chain_route_launches'))"
  head -9 $R/prof_b$b/b${b}_kernel_stats.csv | cut -c1-140; rm -f $R/prof_b$b/*kernel_trace.csv
done
