export TMPDIR=/tmp
rm -rf /tmp/tr; (cd /tmp && timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -o t -- python $GRAFT_REPO_ROOT/bench.py --batch 16 --steps 16 --warmup 2 --no-cpu-baseline --no-prefill > /dev/null 2>&1)
python tools/trace_summary.py /tmp/tr 12
