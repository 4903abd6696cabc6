#!/bin/bash
# Round 3: where the prefill GEMM's time goes -- qgemm_mfma.hip with parts compiled out (MF_KILL; results wrong, timing only)
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT/gpurun_out; cd $GRAFT_REPO_ROOT
for v in ${LIBS:-default mfk1 mfk2 mfk4 mfk3}; do
  L=$GRAFT_REPO_ROOT/exllamav2_amd/libexl2_hip_$v.so; [ "$v" = default ] && L=$GRAFT_REPO_ROOT/exllamav2_amd/libexl2_hip.so
  EXL2_HIP_LIB=$L timeout -k 10 200 python tools/prefill_bench.py --quick --variants auto --reps 5 2>/dev/null | python -c "
import json,sys
for line in sys.stdin:
    try: d=json.loads(line)
    except Exception: continue
    if 'auto' in d: print('$v', d['k'], d['n'], d['m'], d['auto']['ms'], 'ms', d['auto']['TFLOPs'], 'TFLOP/s')"
done | tee $R/r03_prefill_kill.txt
