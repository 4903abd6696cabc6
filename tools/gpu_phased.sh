# batch-of-16 decode through the phased q_gemm route: parity tests, then the kernel breakdown and the bench lines
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_qmatrix.py tests/test_model.py tests/test_tensor_p.py -x -q -m gpu -k "phased or many_rows or tp_shard" 2>&1 | tail -3
rm -rf /tmp/tr; (cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -o t -- python $GRAFT_REPO_ROOT/bench.py --batch 16 --steps 16 --warmup 2 --no-cpu-baseline --no-prefill > /dev/null 2>&1)
python tools/trace_summary.py /tmp/tr 10
timeout 200 python bench.py --batch 16 --steps 32 --warmup 4 --no-cpu-baseline --no-prefill 2>&1 | tail -1
