#!/bin/bash
# A/B of library builds (EXL2_HIP_LIB) on the headline bench, same box, alternating
export TMPDIR=/tmp
for rep in 1 2; do
for lib in ${LIBS:-libexl2_hip.so libexl2_hip_touch.so}; do
  echo "== $lib (rep $rep)"
  EXL2_HIP_LIB=$GRAFT_REPO_ROOT/exllamav2_amd/$lib timeout -k 10 200 python bench.py --no-prefill --no-cpu-baseline --no-ctx-window --no-parity-check --steps 128 --warmup 16 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d['value'], 'tok/s', d['ms_per_step'], 'ms/step', 'frac', d['roofline']['frac'])"
done
done
