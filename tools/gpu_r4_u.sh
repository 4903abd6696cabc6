#!/bin/bash
export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_dropin_reference.py tests/test_ext_tp.py tests/test_chain.py -m gpu -q 2>&1 | grep -E "^E  |FAILED|passed|failed|warn" | head -12 | cut -c1-300
