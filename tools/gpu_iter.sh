#!/bin/bash
# Quick iteration on one MI355X: targeted tests, headline bench, kernel stats.
mkdir -p gpurun_out
export TMPDIR=/tmp
R=gpurun_out
echo "== pytest"; timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -x ${PYTEST_K:+-k "$PYTEST_K"} > $R/pytest_iter.log 2>&1; echo "rc=$?"; tail -5 $R/pytest_iter.log
echo "== bench"; timeout 900 python bench.py --no-cpu-baseline --no-prefill > $R/bench_iter.json 2> $R/bench_iter.err; echo "rc=$?"; cut -c1-900 $R/bench_iter.json; tail -3 $R/bench_iter.err
echo "== bench ctx 1920"; timeout 900 python bench.py --ctx 1920 --no-cpu-baseline --no-prefill 2>/dev/null | cut -c1-200
echo "== rocprof stats"; cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$R/prof_iter -o it -- python $GRAFT_REPO_ROOT/bench.py --steps 64 --warmup 8 --no-cpu-baseline --no-prefill > $GRAFT_REPO_ROOT/$R/rocprof_iter.log 2>&1; echo "rc=$?"
cd $GRAFT_REPO_ROOT
head -12 $R/prof_iter/it_kernel_stats.csv | cut -c1-160
rm -f $R/prof_iter/*kernel_trace.csv
