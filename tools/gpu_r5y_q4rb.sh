#!/bin/bash
# query rows per workgroup of the Q4 decode attention at a grouped-query shape (70B: 8 rows per kv head), short and long context
cd $GRAFT_REPO_ROOT
F="--model llama2-70b --recipe 2.5bpw --cache q4"
TAG=r05y STAGES="ab" AB_NAME=q4_rb REPS=1 AB_STEPS=32 AB_FLAGS="$F" VARIANTS="head rb2=EXL2_Q4_RB=2 rb1=EXL2_Q4_RB=1" tools/gpu_run.sh
TAG=r05y STAGES="ab" AB_NAME=q4_rb_ctx1920 REPS=1 AB_STEPS=32 AB_FLAGS="$F --ctx 1920" VARIANTS="head rb2=EXL2_Q4_RB=2 rb1=EXL2_Q4_RB=1" tools/gpu_run.sh
