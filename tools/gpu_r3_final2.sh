#!/bin/bash
# Round 3, after the prefill change: prefill parity tests, model tests, the default bench line (final r03_bench.json)
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT/gpurun_out; cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -q --timeout 600 > $R/r03_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $R/r03_pytest_gpu.log
timeout -k 10 600 python bench.py > $R/r03_bench.json 2> $R/r03_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r03_bench.json").read().strip().splitlines()[-1])
print(d["value"], "tok/s frac", d["roofline"]["frac"], "traffic", d["roofline"]["traffic"], "prefill", d.get("prefill", {}).get("native"), d.get("prefill", {}).get("roofline"), d.get("prefill", {}).get("parity_check"))
PY
