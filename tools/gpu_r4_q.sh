#!/bin/bash
# tiled hand-off gate|up -> down (EXL2_MLP_TILED: 0 off, 1 where the rows do not fit, 2 from 5 rows up)
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $R; cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_chain.py -m gpu -q 2>&1 | grep -E "^E  .*Assert|FAILED|passed|failed|warn" | head -8 | cut -c1-300
for b in 16 12 8 5; do
  for x in 0 1 2; do
    echo -n "bs=$b TILED=$x : "; EXL2_MLP_TILED=$x timeout -k 10 200 python bench.py --batch $b --steps 32 --warmup 4 --headline-only --no-parity-check 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], 'tok/s', d['ms_per_step'], 'ms')"
  done
done 2>&1 | tee $R/r04_tiled_sweep.txt
echo "== rocprof bs=16"
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/prof_xmem -o xm -- python $GRAFT_REPO_ROOT/bench.py --batch 16 --steps 32 --warmup 4 --headline-only --no-parity-check > /dev/null 2>&1); echo "rc=$?"
head -6 $R/prof_xmem/xm_kernel_stats.csv | cut -c1-150; cp $R/prof_xmem/xm_kernel_stats.csv $R/r04_b16_tiled_kernel_stats.csv; rm -rf $R/prof_xmem
