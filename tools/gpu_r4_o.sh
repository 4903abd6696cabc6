#!/bin/bash
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $R
cd $GRAFT_REPO_ROOT
echo "== overlapped test alone"; timeout -k 10 300 python -m pytest tests/test_chain.py -m gpu -x -q -k "overlapped_chain_equals" 2>&1 | grep -E "^E  .*Assert|FAILED|passed|failed" | head -8 | cut -c1-300
echo "== again, 3 times"; for i in 1 2 3; do timeout -k 10 300 python -m pytest tests/test_chain.py -m gpu -q -k "overlapped_chain_equals" 2>&1 | tail -1; done
echo "== with XMEM tests first"; timeout -k 10 300 python -m pytest tests/test_chain.py -m gpu -q -k "overlapped_chain_equals or operands_from_memory or do_not_fit" 2>&1 | tail -2
echo "== whole file minus overlapped"; timeout -k 10 300 python -m pytest tests/test_chain.py -m gpu -q -k "not overlapped_chain_equals" 2>&1 | tail -2
echo "== rocprof bs=16"
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/prof_xmem -o xm -- python $GRAFT_REPO_ROOT/bench.py --batch 16 --steps 32 --warmup 4 --headline-only --no-parity-check > /dev/null 2>&1); echo "rc=$?"
head -9 $R/prof_xmem/xm_kernel_stats.csv | cut -c1-150; cp $R/prof_xmem/xm_kernel_stats.csv $R/r04_b16_xmem_kernel_stats.csv; rm -rf $R/prof_xmem
