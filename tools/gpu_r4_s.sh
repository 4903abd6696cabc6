#!/bin/bash
# 70B 2.5 bpw + Q4 cache: why does the run WITH the parity check report 57 tok/s and the one without 80?
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $R; cd $GRAFT_REPO_ROOT
B="python bench.py --model llama2-70b --recipe 2.5bpw --cache q4 --steps 32 --warmup 4 --headline-only"
P='import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], "tok/s", d["roofline"]["avg_launch_us"], "us/launch", d["extra"])'
echo -n "no parity      : "; timeout -k 10 300 $B --no-parity-check 2>/dev/null | tail -1 | python -c "$P"
echo -n "parity         : "; timeout -k 10 300 $B 2>/dev/null | tail -1 | python -c "$P"
echo -n "parity ROWS=0  : "; EXL2_LEAN_ROWS=0 timeout -k 10 300 $B 2>/dev/null | tail -1 | python -c "$P"
echo -n "parity PIPE=0  : "; EXL2_LEAN_PIPE=0 timeout -k 10 300 $B 2>/dev/null | tail -1 | python -c "$P"
echo -n "parity, callb  : "; EXL2_HIP_LIB=$GRAFT_REPO_ROOT/exllamav2_amd/libexl2_hip_callb.so timeout -k 10 300 $B 2>/dev/null | tail -1 | python -c "$P"
echo -n "parity, r3     : "; EXL2_HIP_LIB=$GRAFT_REPO_ROOT/exllamav2_amd/libexl2_hip_r3.so timeout -k 10 300 $B 2>/dev/null | tail -1 | python -c "$P"
