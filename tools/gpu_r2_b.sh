#!/bin/bash
# round 2, call B: chained decode on the GPU: parity tests, A/B bench (EXL2_CHAIN=1/0, ring sip 4/0), in-kernel timeline
mkdir -p gpurun_out; export TMPDIR=/tmp; R=gpurun_out
echo "== pytest chain"; timeout 600 python -m pytest tests/test_chain.py tests/test_model.py -m gpu -q --timeout 600 -x > $R/r2b_pytest_chain.log 2>&1; echo "rc=$?"; tail -5 $R/r2b_pytest_chain.log
for v in "EXL2_CHAIN=1" "EXL2_CHAIN=1 EXL2_FLAT_DEPTH=4" "EXL2_CHAIN=0"; do
  echo "== bench $v"
  env $v timeout 300 python bench.py --no-prefill --no-cpu-baseline --steps 128 --warmup 16 2>$R/r2b_bench.err | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d.get('roofline',{})
print(d['value'], 'tok/s', d['ms_per_step'], 'ms/step; gemv avg us', r.get('avg_launch_us'), 'frac', r.get('frac'), 'parity', d.get('parity_check',{}).get('worst_err_over_tol'))"
  tail -2 $R/r2b_bench.err
done
echo "== trace"; timeout 300 python tools/trace_flat.py > $R/r2b_trace_flat.txt 2>&1; echo "rc=$?"; cat $R/r2b_trace_flat.txt | head -80
