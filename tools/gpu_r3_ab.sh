#!/bin/bash
# Round 3: same-box A/B of decode settings.  VARIANTS = space-separated comma-joined env assignments.
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
if [ -z "$SKIP_TESTS" ]; then echo "== pytest"; timeout -k 10 300 python -m pytest tests/test_chain.py tests/test_model.py tests/test_ops.py tests/test_qattn_op.py -m gpu -q --timeout 200 -x > $R/r03ab_pytest.log 2>&1; echo "rc=$?"; tail -3 $R/r03ab_pytest.log; fi
B="python bench.py --steps 64 --warmup 8 --no-cpu-baseline --no-prefill --no-ctx-window --no-parity-check"
i=0
for v in $VARIANTS; do
  i=$((i+1))
  env $(echo $v | tr ',' ' ') timeout -k 10 200 $B > $R/r03ab_bench_$i.json 2>$R/r03ab_bench_$i.err; python - "$R/r03ab_bench_$i.json" "$v" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print("%-40s" % sys.argv[2], d["value"], "tok/s", "avg launch us", d["roofline"]["avg_launch_us"], "frac", d["roofline"]["frac"])
except Exception as e: print(sys.argv[2], "no json", e)
PY
done
if [ -n "$STATS" ]; then
  (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/prof_ab -o ab -- python $GRAFT_REPO_ROOT/bench.py --steps 64 --warmup 8 --no-cpu-baseline --no-prefill --no-parity-check --no-ctx-window > $R/r03ab_stats.log 2>&1); echo "stats rc=$?"
  head -7 $R/prof_ab/ab_kernel_stats.csv | cut -c1-150; rm -f $R/prof_ab/*kernel_trace.csv
fi
