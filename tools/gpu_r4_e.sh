#!/bin/bash
# Round 4, call E: the overlapped chain on the lean kernel (two streams, two graphs, a gate ahead of every launch, sharded counters)
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT/gpurun_out; cd $GRAFT_REPO_ROOT
echo "== pytest (chain)"; timeout -k 10 600 python -m pytest tests/test_chain.py -m gpu -q -x --timeout 300 2>&1 | tail -4
B="--steps 64 --warmup 8 --headline-only --no-parity-check"
for rep in 1 2; do
  for v in "A=1" "EXL2_CHAIN_OVERLAP=1"; do
    echo -n "$v : "; env $v timeout -k 10 200 python bench.py $B 2>$R/r04e_err.txt | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read()); print(d['value'], 'tok/s', d['ms_per_step'], 'ms', d['roofline']['avg_launch_us'], 'us/launch frac', d['roofline']['frac'], d.get('extra',{}).get('chain_overlap'))
except Exception as e: print('failed', e)"; tail -2 $R/r04e_err.txt | grep -v amdgpu
  done
done 2>&1 | tee $R/r04e_overlap_ab.txt
echo "== with the parity check (overlap)"; EXL2_CHAIN_OVERLAP=1 timeout -k 10 300 python bench.py --steps 128 --warmup 16 --headline-only > $R/r04e_bench_overlap.json 2> $R/r04e_bench_overlap.err; echo "rc=$?"; cut -c1-300 $R/r04e_bench_overlap.json; tail -2 $R/r04e_bench_overlap.err | grep -v amdgpu
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r04e_bench_overlap.json").read().strip().splitlines()[-1]); print(d["value"], d["roofline"]["frac"], d.get("parity_check"), d.get("extra"))
except Exception as e: print("parse", e)
PY
