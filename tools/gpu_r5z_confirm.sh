#!/bin/bash
# last call of round 5: the GPU suite on the final source + the 70B line with the Q4 attention's row blocking by capacity
cd $GRAFT_REPO_ROOT
TAG=r05z STAGES="tests" TESTS_TAIL=3 tools/gpu_run.sh
TAG=r05z STAGES="configs" CONFIGS="70b_q4" tools/gpu_run.sh
