#!/bin/bash
# Q4 KV cache decode: attention straight from the codes vs the reference's unpack-per-layer route vs the FP16 cache
for ctx in 0 1920; do
  for v in "CACHE=fp16" "CACHE=q4" "CACHE=q4 EXL2_Q4_FUSED=0"; do
    c=$(echo $v | sed 's/CACHE=\([a-z0-9]*\).*/\1/')
    echo -n "ctx=$ctx $v: "; env $v timeout 600 python bench.py --ctx $ctx --cache $c --no-cpu-baseline --no-prefill 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" || echo failed
  done
done
