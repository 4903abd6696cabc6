#!/bin/bash
# configs[3] on a FRESH box (the end-of-round script runs it as the 13th process on its box and measures 57 tok/s there: same command, same source)
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $R; cd $GRAFT_REPO_ROOT
timeout -k 10 300 python bench.py --model llama2-70b --recipe 2.5bpw --cache q4 --steps 32 --warmup 4 --headline-only 2>/dev/null | tail -1 > $R/r04_bench_70b_q4_fresh_box.json; cut -c1-200 $R/r04_bench_70b_q4_fresh_box.json
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/prof_70b -o p70 -- python $GRAFT_REPO_ROOT/bench.py --model llama2-70b --recipe 2.5bpw --cache q4 --steps 32 --warmup 4 --headline-only --no-parity-check > /dev/null 2>&1); echo "rc=$?"
head -8 $R/prof_70b/p70_kernel_stats.csv | cut -c1-170; cp $R/prof_70b/p70_kernel_stats.csv $R/r04_70b_kernel_stats.csv; rm -rf $R/prof_70b
