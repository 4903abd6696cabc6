#!/bin/bash
# Overlapped chain (EXL2_CHAIN_OVERLAP=1, two graphs side by side) against the serial chain on the same box
export TMPDIR=/tmp
mkdir -p gpurun_out
for ov in 1 0; do
  echo "== EXL2_CHAIN_OVERLAP=$ov"
  EXL2_CHAIN_OVERLAP=$ov timeout -k 10 240 python bench.py --no-prefill --no-cpu-baseline --steps 128 --warmup 16 2> gpurun_out/e_bench_$ov.err | tee gpurun_out/e_bench_$ov.json | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d['value'], 'tok/s', d['ms_per_step'], 'ms/step', 'roofline', d['roofline'].get('frac'), d.get('extra'))"
  tail -2 gpurun_out/e_bench_$ov.err
done
