#!/bin/bash
# Round 4, call G: the ROWS form of the lean kernel (5..16 sequences) on the MI355X: parity, batch sweep against the round-3 routes
# (EXL2_LEAN_ROWS=0), rocprof of bs=16; 70B chain route; drop-in after the host-layer trims.
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT/gpurun_out; cd $GRAFT_REPO_ROOT
echo "== pytest gpu"; timeout -k 10 900 python -m pytest tests -m gpu -q -x --timeout 600 > $R/r04h_pytest_gpu.log 2>&1; echo "rc=$?"; tail -4 $R/r04h_pytest_gpu.log
for b in 5 7 8 12 16; do
  for v in "A=1" "EXL2_LEAN_ROWS=0"; do
    echo -n "batch $b $v: "; env $v timeout -k 10 200 python bench.py --batch $b --steps 32 --warmup 4 --headline-only --no-parity-check 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], 'tok/s', d['ms_per_step'], 'ms')"
  done
done 2>&1 | tee $R/r04h_rows_sweep.txt
echo "== rocprof bs=16"
(cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/prof_b16 -o b16 -- python $GRAFT_REPO_ROOT/bench.py --batch 16 --steps 32 --warmup 4 --headline-only --no-parity-check > $R/r04h_b16.log 2>&1); grep "^{" $R/r04h_b16.log | cut -c1-160
head -8 $R/prof_b16/b16_kernel_stats.csv | cut -c1-150; cp $R/prof_b16/b16_kernel_stats.csv $R/r04h_b16_kernel_stats.csv; rm -rf $R/prof_b16
echo "== default bs=1 (regression check)"
for rep in 1 2; do timeout -k 10 200 python bench.py --steps 64 --warmup 8 --headline-only --no-parity-check 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], 'tok/s', d['roofline']['avg_launch_us'], 'us/launch')"; done
