#!/bin/bash
# tiled xp buffers (EXL2_XP_TILED: 1 default, 0 = ROWS forms for the K = hidden launches)
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $R; cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_chain.py tests/test_model.py -m gpu -q 2>&1 | grep -E "^E  .*Assert|FAILED|passed|failed|warn" | head -8 | cut -c1-300
for b in 16 12 8 5; do
  for x in 0 1; do
    echo -n "bs=$b XP_TILED=$x : "; EXL2_XP_TILED=$x timeout -k 10 200 python bench.py --batch $b --steps 32 --warmup 4 --headline-only 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], 'tok/s', d['ms_per_step'], 'ms', d.get('parity'))" | cut -c1-300
  done
done 2>&1 | tee $R/r04_xp_tiled_sweep.txt
echo "== rocprof bs=16"
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/prof_xmem -o xm -- python $GRAFT_REPO_ROOT/bench.py --batch 16 --steps 32 --warmup 4 --headline-only --no-parity-check > /dev/null 2>&1); echo "rc=$?"
head -8 $R/prof_xmem/xm_kernel_stats.csv | cut -c1-150; cp $R/prof_xmem/xm_kernel_stats.csv $R/r04_b16_xp_tiled_kernel_stats.csv; rm -rf $R/prof_xmem
