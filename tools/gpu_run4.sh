#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== smoke" ; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/smoke.log
echo "== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -q --timeout 600 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/pytest_gpu.log
echo "== microbench"; timeout 900 python tools/microbench.py > gpurun_out/microbench.log 2>&1; echo "micro rc=$?"; cat gpurun_out/microbench.log | grep -v amdgpu.ids | cut -c1-200
echo "== microbench split sweep"; for s in 1 2 4 8 16; do EXL2_GEMV_SPLIT=$s timeout 300 python tools/microbench.py --quick 2>&1 | grep -v "amdgpu.ids\|copy" | cut -c1-160 | sed "s/^/S$s /"; done > gpurun_out/microbench_split.log; cat gpurun_out/microbench_split.log
echo "== bench"; timeout 900 python bench.py --steps 64 --warmup 8 > gpurun_out/bench1.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/bench1.log | cut -c1-1200
echo "== rocprof"; cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof4 -o run4 -- python $GRAFT_REPO_ROOT/bench.py --steps 32 --warmup 4 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/rocprof4.log 2>&1; echo "rocprof rc=$?"
cd $GRAFT_REPO_ROOT; f=$(find gpurun_out/prof4 -name "*kernel_stats*" | head -1); [ -n "$f" ] && head -12 "$f" | cut -c1-160
rm -f gpurun_out/prof4/*kernel_trace.csv 2>/dev/null
