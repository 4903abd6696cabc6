#!/bin/bash
# Round 4, call I: same-box A/B of the lean kernel's states: final source, the first buffer-form pipelined state (commit 549e139), round 3
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT/gpurun_out; cd $GRAFT_REPO_ROOT
B="--steps 64 --warmup 8 --headline-only --no-parity-check"
for rep in 1 2 3; do
  for v in "A=1" "EXL2_HIP_LIB=$GRAFT_REPO_ROOT/exllamav2_amd/libexl2_hip_callb.so" "EXL2_HIP_LIB=$GRAFT_REPO_ROOT/exllamav2_amd/libexl2_hip_r3.so"; do
    echo -n "$(echo $v | sed 's#.*/##') : "; env $v timeout -k 10 200 python bench.py $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], 'tok/s', d['ms_per_step'], 'ms', d['roofline']['avg_launch_us'], 'us/launch frac', d['roofline']['frac'])"
  done
done 2>&1 | tee $R/r04j_states_ab.txt
