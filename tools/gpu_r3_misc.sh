#!/bin/bash
# Round 3: parity tests after the Q4-chain / small-batch / TP changes + small-batch and Q4-cache decode lines
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT/gpurun_out; cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_chain.py tests/test_model.py tests/test_ops.py tests/test_ext_tp.py tests/test_qattn_op.py tests/test_sampling.py -m gpu -q --timeout 300 > $R/r03misc_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $R/r03misc_pytest.log
B="python bench.py --steps 32 --warmup 4 --no-cpu-baseline --no-prefill --no-parity-check --no-ctx-window"
for b in 2 3 4; do timeout -k 10 300 $B --batch $b 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('bs', $b, d['value'], 'tok/s', d['ms_per_step'], 'ms/step', d.get('extra',{}).get('chain_route_launches'))"; done
for c in 1 0; do EXL2_CHAIN=$c timeout -k 10 300 $B --cache q4 --steps 64 --warmup 8 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('7B q4 cache chain=$c', d['value'], 'tok/s', d.get('extra',{}).get('chain_route_launches'))"; done
