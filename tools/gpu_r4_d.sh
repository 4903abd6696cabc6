#!/bin/bash
# Round 4, call D: the ROWS form of the lean kernel (5..16 sequences) on the MI355X: parity, batch sweep against the round-3 routes
# (EXL2_LEAN_ROWS=0), rocprof of bs=16; 70B chain route; drop-in after the host-layer trims.
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT/gpurun_out; cd $GRAFT_REPO_ROOT
echo "== pytest gpu"; timeout -k 10 900 python -m pytest tests -m gpu -q -x --timeout 600 > $R/r04d_pytest_gpu.log 2>&1; echo "rc=$?"; tail -4 $R/r04d_pytest_gpu.log
for b in 5 6 8 12 16; do
  for v in "A=1" "EXL2_LEAN_ROWS=0"; do
    echo -n "batch $b $v: "; env $v timeout -k 10 200 python bench.py --batch $b --steps 32 --warmup 4 --headline-only --no-parity-check 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], 'tok/s', d['ms_per_step'], 'ms')"
  done
done 2>&1 | tee $R/r04d_rows_sweep.txt
echo "== rocprof bs=16"
(cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/prof_b16 -o b16 -- python $GRAFT_REPO_ROOT/bench.py --batch 16 --steps 32 --warmup 4 --headline-only --no-parity-check > $R/r04d_b16.log 2>&1); grep "^{" $R/r04d_b16.log | cut -c1-160
head -8 $R/prof_b16/b16_kernel_stats.csv | cut -c1-150; cp $R/prof_b16/b16_kernel_stats.csv $R/r04d_b16_kernel_stats.csv; rm -rf $R/prof_b16
echo "== 70B 2.5bpw + Q4 cache"
timeout -k 10 400 python bench.py --model llama2-70b --recipe 2.5bpw --cache q4 --steps 32 --warmup 4 --headline-only 2>$R/r04d_70b.err | tail -1 > $R/r04d_bench_70b_q4.json; tail -2 $R/r04d_70b.err
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r04d_bench_70b_q4.json").read().strip().splitlines()[-1]); print("70b", d["value"], d["roofline"]["frac"], d["roofline"]["avg_launch_us"], d.get("parity_check", {}).get("decode_route"), d.get("extra"))
except Exception as e: print("70b parse", e)
PY
echo "== mixtral bs=16 / bs=1"
for b in 1 16; do timeout -k 10 300 python bench.py --model mixtral-8x7b --recipe 3.5bpw --batch $b --steps 32 --warmup 4 --headline-only 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('mixtral bs', $b, d['value'], 'tok/s', d['ms_per_step'], 'ms')"; done 2>&1 | tee -a $R/r04d_rows_sweep.txt
echo "== drop-in decode loop"
timeout -k 10 300 python tools/dropin_decode_bench.py --tokens 128 --attn flash > $R/r04d_dropin_flash.json 2>/dev/null; tail -1 $R/r04d_dropin_flash.json | cut -c1-200
