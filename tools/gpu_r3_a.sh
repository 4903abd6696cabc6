#!/bin/bash
# Round 3, GPU call A: structure probes before any kernel is rewritten + a same-box baseline.
#   lean_probe    : skeleton decode q_gemm -- workgroup size, x staging, combine form (tools/probes/lean_probe.hip)
#   mall_probe    : does reading a launch's weights a few us early help (memory-side cache)
#   persist_probe : one persistent launch with next-phase LDS prefetch vs four launches
#   bench.py      : decode line of the round-2 kernels on this box
#   attn_prefill_bench : the shipped flash-prefill kernel vs torch SDPA (reconcile the numbers)
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
rocm-smi --showproductname 2>/dev/null | grep -i "card series\|gfx" | head -3 > $R/r03_gpu.txt
echo "== lean probe"; timeout -k 5 150 tools/probes/lean_probe > $R/r03_lean_probe.txt 2>&1; echo "rc=$?"; cat $R/r03_lean_probe.txt
echo "== memory-side cache probe"; timeout -k 5 90 tools/probes/mall_probe > $R/r03_mall_probe.txt 2>&1; echo "rc=$?"; cat $R/r03_mall_probe.txt
echo "== persistent-kernel upper bound"; timeout -k 5 60 tools/probes/persist_probe > $R/r03_persist_probe.txt 2>&1; echo "rc=$?"; cat $R/r03_persist_probe.txt
timeout -k 5 100 tools/probes/chain_probe 2>&1 | head -5 >> $R/r03_persist_probe.txt; tail -5 $R/r03_persist_probe.txt
echo "== bench (round-2 kernels, this box)"; timeout -k 10 300 python bench.py --steps 64 --warmup 8 --no-cpu-baseline --no-prefill --no-parity-check --no-ctx-window > $R/r03a_bench.json 2> $R/r03a_bench.err; echo "rc=$?"; cut -c1-400 $R/r03a_bench.json
echo "== flash-prefill attention bench"; timeout -k 10 120 python tools/attn_prefill_bench.py > $R/r03_attn_prefill_bench.jsonl 2> $R/r03_attn_prefill_bench.err; echo "rc=$?"; cat $R/r03_attn_prefill_bench.jsonl | cut -c1-300
