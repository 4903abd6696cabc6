#!/bin/bash
# epilogue inputs in two round trips (was one per dependent load): same-box A/B against commit 549e139's and round 3's lean kernel
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $R; cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  for v in head callb r3; do
    if [ $v = head ]; then E="A=1"; else E="EXL2_HIP_LIB=$GRAFT_REPO_ROOT/exllamav2_amd/libexl2_hip_$v.so"; fi
    [ $v = head ] || [ -f exllamav2_amd/libexl2_hip_$v.so ] || continue
    echo -n "$v : "; env $E timeout -k 10 200 python bench.py --steps 64 --warmup 8 --headline-only --no-parity-check 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], 'tok/s', d['roofline']['avg_launch_us'], 'us/launch', d['roofline']['frac'])"
  done
done 2>&1 | tee $R/r04_states_ab_epilogue.txt
for b in 16 8 4; do timeout -k 10 200 python bench.py --batch $b --steps 32 --warmup 4 --headline-only --no-parity-check 2>/dev/null | tail -1 | cut -c1-160; done | tee -a $R/r04_states_ab_epilogue.txt
timeout -k 10 300 python bench.py --model llama2-70b --recipe 2.5bpw --cache q4 --steps 32 --warmup 4 --headline-only --no-parity-check 2>/dev/null | tail -1 | cut -c1-160 | tee -a $R/r04_states_ab_epilogue.txt
timeout -k 10 600 python -m pytest tests/test_chain.py tests/test_qmatrix.py tests/test_model.py -m gpu -q 2>&1 | tail -2
