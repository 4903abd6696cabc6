#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== smoke" ; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/smoke.log
echo "== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -q --timeout 600 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/pytest_gpu.log
echo "== microbench"; timeout 900 python tools/microbench.py > gpurun_out/microbench.log 2>&1; echo "micro rc=$?"; cat gpurun_out/microbench.log | grep -v amdgpu.ids | cut -c1-200
echo "== microbench waves"; for w in 4 16; do EXL2_GEMV_WAVES=$w timeout 300 python tools/microbench.py --quick 2>&1 | grep -v amdgpu.ids | cut -c1-160 | sed "s/^/w$w /"; done > gpurun_out/microbench_waves.log; cat gpurun_out/microbench_waves.log
echo "== bench"; timeout 900 python bench.py --steps 64 --warmup 8 > gpurun_out/bench1.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/bench1.log | cut -c1-1200
echo "== rocprof"; cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof3 -o run3 -- python $GRAFT_REPO_ROOT/bench.py --steps 32 --warmup 4 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/rocprof3.log 2>&1; echo "rocprof rc=$?"
cd $GRAFT_REPO_ROOT; ls -R gpurun_out/prof3 | head; f=$(find gpurun_out/prof3 -name "*kernel_stats*" | head -1); [ -n "$f" ] && head -20 "$f" | cut -c1-220
rm -f gpurun_out/prof3/*kernel_trace.csv gpurun_out/prof3/*/*kernel_trace.csv 2>/dev/null
