#!/bin/bash
# 70B 2.5 bpw + Q4 cache: the final source against commit 549e139's and round 3's lean kernel on one box
R=gpurun_out; mkdir -p $R
for rep in 1 2; do
  for v in head callb r3; do
    if [ $v = head ]; then E="A=1"; else E="EXL2_HIP_LIB=$GRAFT_REPO_ROOT/exllamav2_amd/libexl2_hip_$v.so"; fi
    [ $v = head ] || [ -f exllamav2_amd/libexl2_hip_$v.so ] || continue
    [ $rep = 2 ] && [ $v = r3 ] && continue
    echo -n "$v : "; env $E timeout -k 10 300 python bench.py --model llama2-70b --recipe 2.5bpw --cache q4 --steps 32 --warmup 4 --headline-only --no-parity-check 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], 'tok/s', d['roofline']['avg_launch_us'], 'us/launch', d['roofline']['frac'])"
  done
done 2>&1 | tee $R/r04_states_ab_70b.txt
