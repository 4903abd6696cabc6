#!/bin/bash
# A/B on the same box: env-var variants of the headline bench (+ optional trace / microbench), interleaved.
mkdir -p gpurun_out
echo "== pytest"; timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 > gpurun_out/pytest_ab.log 2>&1; echo "rc=$?"; tail -3 gpurun_out/pytest_ab.log
for rep in 1 2; do
for v in ${VARIANTS:-"A=1"}; do
  for ctx in ${CTXS:-0}; do
    echo -n "$v ctx=$ctx: "; env $v timeout 600 python bench.py --ctx $ctx --no-cpu-baseline --no-prefill 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'])"
  done
done
done
if [ -n "$TRACE" ]; then timeout 300 python tools/trace_gemv.py 2>&1 | grep -v amdgpu.ids; fi
if [ -n "$MICRO" ]; then for v in ${VARIANTS:-"A=1"}; do echo "== microbench $v"; env $v timeout 600 python tools/microbench.py 2>&1 | grep -v amdgpu.ids | cut -c1-150; done; fi
