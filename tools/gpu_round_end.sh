#!/bin/bash
# Round-end validation sized for ~2 GPU-minutes: smoke, parity tests, headline bench line (roofline + cpu_baseline +
# prefill), rocprofv3 kernel stats of the same decode loop, batch-16 line.  Outputs -> gpurun_out/ (copied to profiles/).
mkdir -p gpurun_out
export TMPDIR=/tmp
R=gpurun_out
echo "== smoke"; timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $R/smoke.log 2>&1; echo "rc=$?"; tail -1 $R/smoke.log
echo "== pytest gpu"; timeout 300 python -m pytest tests -m gpu -q --timeout 300 > $R/pytest_gpu.log 2>&1; echo "rc=$?"; tail -3 $R/pytest_gpu.log
echo "== bench"; timeout 300 python bench.py > $R/bench_r01.json 2> $R/bench_r01.err; echo "rc=$?"; cut -c1-1600 $R/bench_r01.json
echo "== rocprof stats (bs=1)"; (cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$R/prof_stats -o r01 -- python $GRAFT_REPO_ROOT/bench.py --steps 64 --warmup 8 --no-cpu-baseline --no-prefill > $GRAFT_REPO_ROOT/$R/rocprof_stats.log 2>&1); echo "rc=$?"
head -8 $R/prof_stats/r01_kernel_stats.csv | cut -c1-150
rm -f $R/prof_stats/*kernel_trace.csv
echo "== batch 16"
timeout 120 python bench.py --batch 16 --steps 32 --warmup 4 --no-cpu-baseline --no-prefill 2>/dev/null | tail -1 | cut -c1-260 | tee $R/bench_b16.json
rocm-smi --showproductname 2>/dev/null | grep -i "card series\|gfx" | head -3 > $R/gpu.txt
