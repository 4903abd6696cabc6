TAG=r10e STAGES="tests" TESTS="tests/test_ops.py tests/test_model.py tests/test_dropin_reference.py" TESTS_TIMEOUT=1200 tools/gpu_run.sh
for v in "" "EXL2_ROPE_ROWS_MIN=100000000"; do
echo "== bench prefill $v"; env $v python bench.py --steps 20 --warmup 5 --no-parity-check --no-dropin --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d.get('prefill') or d.get('extra',{}).get('prefill'); print(d['value']); print(json.dumps(p)[:900])"
done
