#!/bin/bash
# Round 3: 5..16 sequences as row groups of 4 on the round-3 kernel (EXL2_CHAIN_ROWGROUPS=1) vs the round-2 kernels (=0)
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT/gpurun_out; cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_chain.py -m gpu -q --timeout 300 -k "many_rows or equals_oracle" > $R/r03g_pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $R/r03g_pytest.log
B="python bench.py --steps 32 --warmup 4 --no-cpu-baseline --no-prefill --no-parity-check --no-ctx-window"
for b in ${BATCHES:-6 8 12 16}; do for g in 1 0; do EXL2_CHAIN_ROWGROUPS=$g timeout -k 10 300 $B --batch $b 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('bs', $b, 'groups=$g', d['value'], 'tok/s', d['ms_per_step'], 'ms/step', d.get('extra',{}).get('chain_route_launches'))"; done; done | tee $R/r03_rowgroups.txt
