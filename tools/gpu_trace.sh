#!/bin/bash
# kernel trace of the headline bench, summarised per launch shape
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT/gpurun_out
for v in ${VARIANTS:-"A=1"}; do
  echo "=== $v"
  env $v timeout 600 python bench.py --no-cpu-baseline --no-prefill 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'])"
  rm -rf /tmp/tr; (cd /tmp && env $v timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 32 --warmup 4 --no-cpu-baseline --no-prefill > /dev/null 2>&1)
  python tools/trace_summary.py /tmp/tr 12
done
