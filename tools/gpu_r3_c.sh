#!/bin/bash
# Round 3, GPU call C: in-kernel timeline of the lean kernel (EXL2_TRACE build) + the decode line at the three register budgets.
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
echo "== pytest gpu (chain)"; timeout -k 10 300 python -m pytest tests/test_chain.py -m gpu -q --timeout 200 -x > $R/r03c_pytest.log 2>&1; echo "rc=$?"; tail -3 $R/r03c_pytest.log
for occ in ${TRACE_OCCS:-6}; do
echo "== trace OCC=$occ"; EXL2_LEAN_OCC=$occ timeout -k 10 200 python tools/trace_lean.py > $R/r03c_trace_lean_occ$occ.txt 2>&1; echo "rc=$?"; cat $R/r03c_trace_lean_occ$occ.txt | grep -v "amdgpu.ids"
done
B="python bench.py --steps 64 --warmup 8 --no-cpu-baseline --no-prefill --no-ctx-window --no-parity-check"
for v in "EXL2_LEAN_OCC=4" "EXL2_LEAN_OCC=6" "EXL2_LEAN_OCC=8" $EXTRA_VARIANTS; do
  echo "== bench $v"; env $v timeout -k 10 200 $B > $R/r03c_bench_$v.json 2> $R/r03c_bench_$v.err; echo "rc=$?"; python - "$R/r03c_bench_$v.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(d["value"], "tok/s", d["ms_per_step"], "ms", "avg launch us", d["roofline"]["avg_launch_us"], "frac", d["roofline"]["frac"], d.get("extra"))
except Exception as e: print("no json", e)
PY
done
