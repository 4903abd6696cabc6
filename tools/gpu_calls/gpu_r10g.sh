TAG=r10g STAGES="tests" TESTS="tests/test_chain.py tests/test_ops.py tests/test_model.py tests/test_moe.py" TESTS_TIMEOUT=1200 tools/gpu_run.sh
for v in "EXL2_Q4_LAUNCHES=1" "EXL2_Q4_LAUNCHES=4"; do
echo "== tinyllama gptq + Q4 cache $v"; env $v python bench.py --model tinyllama --recipe gptq-4bit-128g --cache q4 --steps 64 --warmup 8 --headline-only 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['windows']['tokens_per_s'], 'parity', d.get('parity_check',{}).get('worst_err_over_tol'))"
done
