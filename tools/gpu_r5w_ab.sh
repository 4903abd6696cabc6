cd $GRAFT_REPO_ROOT
L=$PWD/exllamav2_amd/libexl2_hip_noring.so
TAG=r05w STAGES="tests" TESTS_TAIL=6 tools/gpu_run.sh
TAG=r05w STAGES="ab" AB_NAME=7b REPS=3 VARIANTS="head noring=EXL2_HIP_LIB=$L" tools/gpu_run.sh
TAG=r05w STAGES="ab" AB_NAME=70b REPS=2 AB_STEPS=32 AB_FLAGS="--model llama2-70b --recipe 2.5bpw --cache q4" VARIANTS="head q4l2=EXL2_Q4_LAUNCHES=2 q4l4=EXL2_Q4_LAUNCHES=4 s8p1=EXL2_LEAN_S8_PASSES=1 ringoff=EXL2_HIP_LIB=$L old=EXL2_HIP_LIB=$L,EXL2_Q4_LAUNCHES=4,EXL2_LEAN_S8_PASSES=1" tools/gpu_run.sh
TAG=r05w STAGES="ab" AB_NAME=7bq4 REPS=2 AB_STEPS=128 AB_FLAGS="--cache q4" VARIANTS="head q4l2=EXL2_Q4_LAUNCHES=2 q4l4=EXL2_Q4_LAUNCHES=4" tools/gpu_run.sh
R=$PWD/gpurun_out
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/prof70 -o r05w70 -- python $GRAFT_REPO_ROOT/bench.py --model llama2-70b --recipe 2.5bpw --cache q4 --steps 32 --warmup 4 --windows 1 --headline-only --no-parity-check > $R/r05w_rocprof_70b.log 2>&1); echo "rc=$?"
f=$(find $R/prof70 -name "r05w70_kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $R/r05w_70b_kernel_stats.csv && head -9 $f | cut -c1-150
rm -rf $R/prof70
