#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_chain.py tests/test_dropin_reference.py tests/test_ext_tp.py -m gpu -q -k "overlapped or tensor_parallel or ext_tp or tp_" 2>&1 | tail -40 | tee gpurun_out/f_pytest.txt
