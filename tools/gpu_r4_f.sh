#!/bin/bash
# Round 4, call F: overlapped chain, second look: longer poll sleep, with / without the per-launch gates
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT/gpurun_out; cd $GRAFT_REPO_ROOT
B="--steps 64 --warmup 8 --headline-only --no-parity-check"
for rep in 1 2; do
  for v in "A=1" "EXL2_CHAIN_OVERLAP=1" "EXL2_CHAIN_OVERLAP=1 EXL2_CHAIN_GATES=0"; do
    echo -n "$v : "; env $v timeout -k 10 200 python bench.py $B 2>$R/r04f_err.txt | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read()); print(d['value'], 'tok/s', d['ms_per_step'], 'ms', d['roofline']['avg_launch_us'], 'us/launch', d.get('extra',{}).get('chain_overlap'))
except Exception as e: print('failed', e)"; tail -2 $R/r04f_err.txt | grep -v amdgpu
  done
done 2>&1 | tee $R/r04f_overlap_ab.txt
