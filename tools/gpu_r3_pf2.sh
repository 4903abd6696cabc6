#!/bin/bash
# Round 3: prefill GEMM with the weights decoded once per call (auto, >= 2048 rows) vs decoded inside the GEMM
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT/gpurun_out; cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_qmatrix.py -m gpu -q --timeout 300 -k prefill > $R/r03pf_pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $R/r03pf_pytest.log
timeout -k 10 400 python tools/prefill_bench.py --variants auto,decode_in_gemm,tile256_mt8 --reps 5 2>/dev/null > $R/r03_prefill_gemm.jsonl
python - <<'PY'
import json
for line in open("gpurun_out/r03_prefill_gemm.jsonl"):
    try: d = json.loads(line)
    except Exception: continue
    if "auto" in d: print(d["k"], d["n"], d["m"], {k: (d[k]["ms"], d[k]["TFLOPs"]) for k in ("auto", "decode_in_gemm", "reconstruct_plus_hipblaslt") if k in d}, "maxdiff", d.get("decode_in_gemm", {}).get("max_abs_diff_vs_auto"))
PY
