#!/bin/bash
# Round 4, call K: which change of the lean kernel cost the 5 %: same-box runs of its states (variant libraries differ in qgemv_lean.o only)
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT/gpurun_out; cd $GRAFT_REPO_ROOT
B="--steps 64 --warmup 8 --headline-only --no-parity-check"
for rep in 1 2; do
  for v in head callb r3; do
    if [ $v = head ]; then E="A=1"; else E="EXL2_HIP_LIB=$GRAFT_REPO_ROOT/exllamav2_amd/libexl2_hip_$v.so"; fi
    echo -n "$v : "; env $E timeout -k 10 200 python bench.py $B 2>$R/r04k_err_$v.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], 'tok/s', d['roofline']['avg_launch_us'], 'us/launch')" || tail -3 $R/r04k_err_$v.txt
  done
done 2>&1 | tee $R/r04m_bisect.txt
