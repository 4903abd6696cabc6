#!/bin/bash
# Round 3, GPU call B: the lean chain kernel (csrc/qgemv_lean.hip) on the MI355X for the first time --
# parity tests that touch it, then same-box A/B of the decode line: round-2 kernel vs lean at three register budgets,
# then rocprofv3 kernel stats of the lean run.
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
echo "== pytest gpu (chain / model / qmatrix)"; timeout -k 10 420 python -m pytest tests/test_chain.py tests/test_model.py tests/test_qmatrix.py -m gpu -q --timeout 300 -x > $R/r03b_pytest.log 2>&1; echo "rc=$?"; tail -5 $R/r03b_pytest.log
B="python bench.py --steps 64 --warmup 8 --no-cpu-baseline --no-prefill --no-ctx-window"
echo "== bench lean (default) with the parity check"; timeout -k 10 300 $B > $R/r03b_bench_lean.json 2> $R/r03b_bench_lean.err; echo "rc=$?"; cut -c1-330 $R/r03b_bench_lean.json; tail -2 $R/r03b_bench_lean.err
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r03b_bench_lean.json").read().strip().splitlines()[-1])
    print("roofline", d.get("roofline")); print("parity", d.get("parity_check")); print("extra", json.dumps(d.get("extra"))[:600])
except Exception as e: print("no json", e)
PY
for v in "EXL2_LEAN=0" "EXL2_LEAN_OCC=4" "EXL2_LEAN_OCC=5" "EXL2_LEAN_OCC=6" "EXL2_LEAN_TPW=2"; do
  echo "== bench $v"; env $v timeout -k 10 200 $B --no-parity-check > $R/r03b_bench_$v.json 2> $R/r03b_bench_$v.err; echo "rc=$?"; python - "$R/r03b_bench_$v.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(d["value"], "tok/s", d["ms_per_step"], "ms", d.get("roofline"))
except Exception as e: print("no json", e)
PY
done
echo "== rocprof stats (lean)"; (cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/prof_stats_b -o r03b -- python $GRAFT_REPO_ROOT/bench.py --steps 64 --warmup 8 --no-cpu-baseline --no-prefill --no-parity-check --no-ctx-window > $R/r03b_rocprof_stats.log 2>&1); echo "rc=$?"
find $R/prof_stats_b -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'head -12 {} | cut -c1-200'
find $R/prof_stats_b -name "*kernel_trace.csv" -delete
