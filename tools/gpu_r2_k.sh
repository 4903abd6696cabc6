#!/bin/bash
# Round 2: the chained decode route inside pipeline stages (first / middle / last) on the GPU, messages passed by hand.
mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT/gpurun_out; cd $GRAFT_REPO_ROOT
timeout -k 3 25 python -m pytest tests/test_pipeline.py -m gpu -q --timeout 20 > $R/r02k_pytest_pipeline.log 2>&1; echo "rc=$?"; tail -5 $R/r02k_pytest_pipeline.log
