TAG=r09f STAGES="tests" TESTS="tests/test_moe.py tests/test_chain.py" TESTS_TIMEOUT=900 tools/gpu_run.sh
for b in 2 4; do for v in "" "EXL2_MOE_NO_LEAN=1"; do
  echo "== mixtral b$b $v"; env $v python bench.py --model mixtral-8x7b --recipe 3.5bpw --batch $b --steps 32 --warmup 4 --headline-only 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['windows']['tokens_per_s'], 'parity', d.get('parity_check',{}).get('worst_err_over_tol'), d.get('parity_check',{}).get('rows_skipped_router_near_tie'))"
done; done
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/prof_mx -o r09f -- python /root/repo/bench.py --model mixtral-8x7b --recipe 3.5bpw --batch 4 --steps 24 --warmup 4 --windows 1 --headline-only --no-parity-check > /root/repo/gpurun_out/r09f_mx_rocprof.log 2>&1)
f=$(find gpurun_out/prof_mx -name "r09f_kernel_stats.csv" | head -1); cp $f gpurun_out/r09f_mixtral_b4_kernel_stats.csv; head -9 $f | cut -c1-200; rm -rf gpurun_out/prof_mx
