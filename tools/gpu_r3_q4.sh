#!/bin/bash
# Round 3: Q4-cache decode on the chained route -- parity tests + configs[3] bench (70B 2.5bpw, Q4 cache), chained vs not.
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT/gpurun_out; cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_chain.py tests/test_model.py tests/test_ops.py tests/test_ext_tp.py tests/test_qattn_op.py -m gpu -q --timeout 300 -x > $R/r03q4_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $R/r03q4_pytest.log
for c in 1 0; do
  EXL2_CHAIN=$c timeout -k 10 400 python bench.py --model llama2-70b --recipe 2.5bpw --cache q4 --steps 32 --warmup 4 --no-cpu-baseline --no-prefill 2>$R/r03q4_70b_$c.err | tail -1 > $R/r03q4_70b_chain$c.json
  python - $R/r03q4_70b_chain$c.json $c <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read()); print("70B 2.5bpw q4 chain=%s" % sys.argv[2], d["value"], "tok/s", d["roofline"].get("frac"), d.get("extra", {}).get("chain_route_launches"))
except Exception as e: print("no json", e)
PY
done
EXL2_CHAIN=1 timeout -k 10 300 python bench.py --cache q4 --steps 64 --warmup 8 --no-cpu-baseline --no-prefill --no-ctx-window 2>/dev/null | tail -1 | cut -c1-200
