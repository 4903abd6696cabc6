#!/bin/bash
# rocprofv3 kernel stats of the 70B 2.5bpw + Q4 cache decode loop on the final source
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; R=$PWD/gpurun_out; export TMPDIR=/tmp
(cd /tmp && timeout -k 5 120 rocprofv3 --kernel-trace --stats --output-format csv -d $R/prof70 -o r05z70 -- python $GRAFT_REPO_ROOT/bench.py --model llama2-70b --recipe 2.5bpw --cache q4 --steps 32 --warmup 4 --windows 1 --headline-only --no-parity-check > $R/r05z_rocprof_70b.log 2>&1); echo "rc=$?"
f=$(find $R/prof70 -name "r05z70_kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $R/r05z_70b_kernel_stats.csv && head -7 $f | cut -c1-150
rm -rf $R/prof70
