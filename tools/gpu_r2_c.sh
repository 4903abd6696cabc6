#!/bin/bash
# round 2, call C: chained decode: tests, bench variants, per-launch-shape kernel durations in graph mode (rocprofv3)
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT/gpurun_out
echo "== pytest chain"; timeout 600 python -m pytest tests/test_chain.py tests/test_model.py -m gpu -q --timeout 600 -x > $R/r2c_pytest_chain.log 2>&1; echo "rc=$?"; tail -3 $R/r2c_pytest_chain.log
for v in ${VARIANTS:-"EXL2_CHAIN=1" "EXL2_CHAIN=1 EXL2_FLAT_DEPTH=6" "EXL2_CHAIN=0"}; do
  echo "== bench $v"
  env $v timeout 300 python bench.py --no-prefill --no-cpu-baseline --steps 128 --warmup 16 2>$R/r2c_bench.err | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d.get('roofline',{})
print(d['value'], 'tok/s', d['ms_per_step'], 'ms/step; gemv avg us', r.get('avg_launch_us'), 'frac', r.get('frac'), 'parity', d.get('parity_check',{}).get('worst_err_over_tol'))"
  tail -2 $R/r2c_bench.err | grep -v amdgpu.ids
done
for v in ${PROF_VARIANTS:-"EXL2_CHAIN=1"}; do
  echo "== rocprof kernel trace $v (graph mode, 32 steps)"
  rm -rf /tmp/tr; (cd /tmp && env $v timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 32 --warmup 4 --no-cpu-baseline --no-prefill --no-parity-check > /dev/null 2>&1)
  python tools/trace_summary.py /tmp/tr 14
done
