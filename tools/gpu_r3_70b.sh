#!/bin/bash
# Round 3: configs[3] (70B 2.5bpw, Q4 cache) on the chained route + parity tests of the touched files
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT/gpurun_out; cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_chain.py tests/test_model.py -m gpu -q --timeout 300 > $R/r03c_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $R/r03c_pytest.log
for c in 1 0; do
  EXL2_CHAIN=$c timeout -k 10 500 python bench.py --model llama2-70b --recipe 2.5bpw --cache q4 --steps 32 --warmup 4 --no-cpu-baseline --no-prefill 2>$R/r03c_70b_$c.err | tail -1 > $R/r03c_70b_chain$c.json
  python - $R/r03c_70b_chain$c.json $c <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read()); print("70B 2.5bpw q4 chain=%s" % sys.argv[2], d["value"], "tok/s", d.get("roofline", {}).get("frac"), d.get("extra", {}).get("chain_route_launches"), d.get("parity_check", {}).get("confident_tokens_equal"))
except Exception as e: print("no json", e)
PY
done
timeout -k 10 300 python bench.py --steps 64 --warmup 8 --no-cpu-baseline --no-prefill --no-parity-check --no-ctx-window 2>/dev/null | tail -1 | cut -c1-150
