#!/bin/bash
# A/B of library builds (EXL2_HIP_LIB) on the headline bench
export TMPDIR=/tmp
for lib in ${LIBS:-libexl2_hip.so libexl2_hip_d6.so libexl2_hip_d8.so}; do
  echo "== $lib"
  EXL2_HIP_LIB=$GRAFT_REPO_ROOT/exllamav2_amd/$lib timeout 300 python bench.py --no-prefill --no-cpu-baseline --steps 128 --warmup 16 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d['value'], 'tok/s', d['ms_per_step'], 'ms/step')"
done
