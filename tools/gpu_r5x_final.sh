#!/bin/bash
# end-of-round evidence of the final source + the A/B of where the one-launch Q4 step packs its rows
cd $GRAFT_REPO_ROOT
TAG=r05x STAGES="smoke tests bench_default stats" TESTS_TAIL=3 tools/gpu_run.sh
TAG=r05x STAGES="configs" CONFIGS="7b_q4cache 70b_q4" tools/gpu_run.sh
TAG=r05x STAGES="ab" AB_NAME=pack_7bq4 REPS=1 AB_STEPS=128 AB_FLAGS="--cache q4" VARIANTS="head first=EXL2_Q4_PACK_FIRST=1 head2" tools/gpu_run.sh
TAG=r05x STAGES="ab" AB_NAME=pack_70b REPS=1 AB_STEPS=32 AB_FLAGS="--model llama2-70b --recipe 2.5bpw --cache q4" VARIANTS="head first=EXL2_Q4_PACK_FIRST=1" tools/gpu_run.sh
TAG=r05x STAGES="bench_driver" tools/gpu_run.sh
