#!/bin/bash
# Round 2: the two sampler tests written after the last full GPU run (odd vocabulary sizes, decoder.run_sampled).
mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT/gpurun_out; cd $GRAFT_REPO_ROOT
EXL2_RUN_UNVERIFIED=1 timeout -k 2 15 python -m pytest tests/test_sampling.py -m gpu -q --timeout 12 -k "off_the_quad or run_sampled" > $R/r02l_pytest.log 2>&1; echo "rc=$?"; tail -4 $R/r02l_pytest.log
