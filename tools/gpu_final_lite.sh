#!/bin/bash
# Short round-end validation (fits a few GPU-minutes): smoke, parity tests, headline bench, kernel stats, batch / MoE lines.
mkdir -p gpurun_out
export TMPDIR=/tmp
R=gpurun_out
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $R/smoke.log 2>&1; echo "rc=$?"; tail -1 $R/smoke.log
echo "== pytest gpu"; timeout 600 python -m pytest tests -m gpu -q --timeout 600 > $R/pytest_gpu.log 2>&1; echo "rc=$?"; tail -3 $R/pytest_gpu.log
echo "== bench"; timeout 400 python bench.py > $R/bench_r01.json 2> $R/bench_r01.err; echo "rc=$?"; cut -c1-1400 $R/bench_r01.json
echo "== batch 16 / 8"
timeout 200 python bench.py --batch 16 --steps 32 --warmup 4 --no-cpu-baseline --no-prefill 2>/dev/null | tail -1 | cut -c1-260 | tee $R/bench_b16.json
echo "== batch 16 kernel breakdown"
rm -rf /tmp/tr; (cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -o t -- python $GRAFT_REPO_ROOT/bench.py --batch 16 --steps 16 --warmup 2 --no-cpu-baseline --no-prefill > /dev/null 2>&1)
python tools/trace_summary.py /tmp/tr 8 | tee $R/batch16_kernels.txt
echo "== rocprof stats (bs=1)"; (cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$R/prof_stats -o r01 -- python $GRAFT_REPO_ROOT/bench.py --steps 64 --warmup 8 --no-cpu-baseline --no-prefill > $GRAFT_REPO_ROOT/$R/rocprof_stats.log 2>&1); echo "rc=$?"
head -8 $R/prof_stats/r01_kernel_stats.csv | cut -c1-150
rm -f $R/prof_stats/*kernel_trace.csv
echo "== mixtral bs=16"
timeout 240 python bench.py --model mixtral-8x7b --recipe 3.5bpw --batch 16 --steps 16 --warmup 2 --no-cpu-baseline --no-prefill 2>/dev/null | tail -1 | cut -c1-260 | tee $R/bench_mixtral_b16.json
