#!/bin/bash
# First GPU pass: smoke, parity tests, microbench, bench, kernel trace.  Everything logs under gpurun_out/.
mkdir -p gpurun_out
export TMPDIR=/tmp
rocminfo | grep -E "Marketing Name|Compute Unit|gfx" | head -8 > gpurun_out/gpu.txt 2>&1
echo "== smoke" ; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -5 gpurun_out/smoke.log
echo "== pytest gpu"; timeout 1200 python -m pytest tests -m gpu -q -x --timeout 600 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/pytest_gpu.log
echo "== microbench"; timeout 900 python tools/microbench.py > gpurun_out/microbench.log 2>&1; echo "micro rc=$?"; cat gpurun_out/microbench.log | tail -20
echo "== bench"; timeout 900 python bench.py --steps 64 --warmup 8 > gpurun_out/bench1.log 2>&1; echo "bench rc=$?"; tail -5 gpurun_out/bench1.log
echo "== rocprof"; cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof1 -o run1 -- python $GRAFT_REPO_ROOT/bench.py --steps 32 --warmup 4 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/rocprof1.log 2>&1; echo "rocprof rc=$?"
cd $GRAFT_REPO_ROOT; find gpurun_out/prof1 -name "*stats*" | head; f=$(find gpurun_out/prof1 -name "*kernel_stats*" | head -1); [ -n "$f" ] && head -20 "$f"
