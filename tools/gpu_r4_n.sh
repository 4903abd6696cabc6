#!/bin/bash
# XMEM form (A operands from memory): where the rows do not fit (default, =1), everywhere from 5 rows (=2), off (=0)
R=gpurun_out; mkdir -p $R
timeout -k 10 300 python -m pytest tests/test_chain.py -m gpu -x -q 2>&1 | tail -2
for b in 16 12 8 5; do
  for x in 0 1 2; do
    echo -n "bs=$b XMEM=$x : "; EXL2_LEAN_XMEM=$x timeout -k 10 200 python bench.py --batch $b --steps 32 --warmup 4 --headline-only --no-parity-check 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], 'tok/s', d['ms_per_step'], 'ms')"
  done
done 2>&1 | tee $R/r04_xmem_sweep.txt
cd /tmp && export TMPDIR=/tmp
for x in 1 2; do
  EXL2_LEAN_XMEM=$x timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$R/prof_xmem$x -o x -- python $GRAFT_REPO_ROOT/bench.py --batch 16 --steps 32 --warmup 4 --headline-only --no-parity-check > /dev/null 2>&1
  f=$(find $GRAFT_REPO_ROOT/$R/prof_xmem$x -name "*kernel_stats.csv" | head -1); head -8 "$f" | cut -c1-150; cp "$f" $GRAFT_REPO_ROOT/$R/r04_b16_xmem${x}_kernel_stats.csv; rm -rf $GRAFT_REPO_ROOT/$R/prof_xmem$x
done
