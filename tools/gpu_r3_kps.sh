#!/bin/bash
# keys per attention split: default bench window (ctx 16..144) and the ctx-1920 datapoint
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT/gpurun_out; cd $GRAFT_REPO_ROOT
for k in 128 256 512 1024; do
  EXL2_ATT_KPS=$k timeout -k 10 300 python bench.py --no-cpu-baseline --no-prefill --no-parity-check > $R/r03kps_$k.json 2>/dev/null
  python - $R/r03kps_$k.json $k <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); e = d.get("extra", {})
print("kps", sys.argv[2], d["value"], "tok/s;", {k: v for k, v in e.items() if "ctx" in k})
PY
done
