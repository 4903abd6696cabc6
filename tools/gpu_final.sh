#!/bin/bash
# Round-end style validation on one MI355X: smoke, parity tests, headline bench, kernel trace, HBM traffic counters.
mkdir -p gpurun_out
export TMPDIR=/tmp
R=gpurun_out
rocminfo | grep -E "Marketing Name|Compute Unit|gfx" | head -6 > $R/gpu.txt 2>&1
echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $R/smoke.log 2>&1; echo "rc=$?"; tail -1 $R/smoke.log
echo "== pytest gpu"; timeout 1800 python -m pytest tests -m gpu -q --timeout 900 > $R/pytest_gpu.log 2>&1; echo "rc=$?"; tail -3 $R/pytest_gpu.log
echo "== bench"; timeout 900 python bench.py > $R/bench_r01.json 2> $R/bench_r01.err; echo "rc=$?"; cat $R/bench_r01.json | cut -c1-1500
echo "== bench ctx 1920"; timeout 900 python bench.py --ctx 1920 --no-cpu-baseline --no-prefill > $R/bench_r01_ctx1920.json 2>/dev/null; echo "rc=$?"; cat $R/bench_r01_ctx1920.json | cut -c1-400
echo "== microbench"; timeout 900 python tools/microbench.py > $R/microbench.log 2>&1; grep -v amdgpu.ids $R/microbench.log | cut -c1-170
echo "== prefill (config 3) + MoE (config 5) + in-kernel timeline"
timeout 600 python tools/prefill_bench.py > $R/prefill_gemm.jsonl 2>/dev/null; cut -c1-200 $R/prefill_gemm.jsonl
timeout 600 python tools/prefill_bench.py --model > $R/prefill_model.json 2>/dev/null; cat $R/prefill_model.json
EXL2_NATIVE_PREFILL=1 timeout 600 python tools/prefill_bench.py --model --layers 4 > $R/prefill_model_native_4layers.json 2>/dev/null; cat $R/prefill_model_native_4layers.json
timeout 600 python tools/moe_bench.py > $R/moe_bench.jsonl 2>/dev/null; cut -c1-200 $R/moe_bench.jsonl
if [ -f exllamav2_amd/libexl2_hip_trace.so ]; then timeout 300 python tools/trace_gemv.py 2>&1 | grep -v amdgpu.ids > $R/trace_mlp.txt; grep "mlp\|span" $R/trace_mlp.txt; fi
timeout 120 ./tools/probes/chain_probe > $R/stream_probe.txt 2>&1; tail -4 $R/stream_probe.txt
echo "== rocprof stats"; cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$R/prof_stats -o r01 -- python $GRAFT_REPO_ROOT/bench.py --steps 64 --warmup 8 --no-cpu-baseline --no-prefill > $GRAFT_REPO_ROOT/$R/rocprof_stats.log 2>&1; echo "rc=$?"
echo "== rocprof pmc FETCH_SIZE"; timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$R/prof_pmc_fetch -o r01 -- python $GRAFT_REPO_ROOT/bench.py --steps 16 --warmup 2 --no-cpu-baseline --no-prefill --no-graph > $GRAFT_REPO_ROOT/$R/rocprof_pmc.log 2>&1; echo "rc=$?"
echo "== rocprof pmc WRITE_SIZE"; timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$R/prof_pmc_write -o r01 -- python $GRAFT_REPO_ROOT/bench.py --steps 16 --warmup 2 --no-cpu-baseline --no-prefill --no-graph > /dev/null 2>&1; echo "rc=$?"
cd $GRAFT_REPO_ROOT
head -12 $R/prof_stats/r01_kernel_stats.csv | cut -c1-150
python - <<'PY'
import csv, glob, collections, json
out = {}
for name in ("fetch", "write"):
    fs = glob.glob(f"gpurun_out/prof_pmc_{name}/*counter_collection.csv")
    if not fs: print(name, "no counter file"); continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        agg[(r["Kernel_Name"][:48], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:8]:
        print(name, k, c, "launches", len(v), "avg", round(sum(v) / len(v), 1))
        out[f"{c}:{k}"] = {"launches": len(v), "avg": sum(v) / len(v)}
json.dump(out, open("gpurun_out/pmc_summary.json", "w"), indent=1)
PY
rm -f $R/prof_stats/*kernel_trace.csv $R/prof_pmc_fetch/*kernel_trace.csv $R/prof_pmc_write/*kernel_trace.csv $R/prof_pmc_*/*counter_collection.csv
