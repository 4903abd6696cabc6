#!/bin/bash
# Round-2 evidence on one MI355X: smoke, all -m gpu parity tests, headline bench (roofline + cpu_baseline + prefill), rocprofv3
# kernel stats of the same decode loop, PMC FETCH_SIZE pass, other-config datapoints.  Outputs -> gpurun_out/ (copied to profiles/).
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
echo "== smoke"; timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $R/r02_smoke.log 2>&1; echo "rc=$?"; tail -1 $R/r02_smoke.log
echo "== pytest gpu"; timeout -k 10 900 python -m pytest tests -m gpu -q --timeout 600 > $R/r02_pytest_gpu.log 2>&1; echo "rc=$?"; tail -3 $R/r02_pytest_gpu.log
echo "== bench"; timeout -k 10 600 python bench.py > $R/r02_bench.json 2> $R/r02_bench.err; echo "rc=$?"; cut -c1-2400 $R/r02_bench.json
echo "== rocprof stats (bs=1)"; (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/prof_stats -o r02 -- python $GRAFT_REPO_ROOT/bench.py --steps 64 --warmup 8 --no-cpu-baseline --no-prefill --no-parity-check --no-ctx-window > $R/r02_rocprof_stats.log 2>&1); echo "rc=$?"
head -8 $R/prof_stats/r02_kernel_stats.csv | cut -c1-160
rm -f $R/prof_stats/*kernel_trace.csv
echo "== rocprof pmc FETCH_SIZE"; (cd /tmp && timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/prof_pmc_fetch -o r02 -- python $GRAFT_REPO_ROOT/bench.py --steps 16 --warmup 2 --no-cpu-baseline --no-prefill --no-graph --no-parity-check --no-ctx-window > $R/r02_rocprof_pmc.log 2>&1); echo "rc=$?"
python - <<'PY'
import csv, glob, collections, json
out = {}
fs = glob.glob("gpurun_out/prof_pmc_fetch/**/*counter_collection.csv", recursive=True)
if fs:
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        agg[(r["Kernel_Name"][:64], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:8]:
        print(k, c, "launches", len(v), "avg", round(sum(v) / len(v), 1))
        out[f"{c}:{k}"] = {"launches": len(v), "avg": sum(v) / len(v)}
    json.dump(out, open("gpurun_out/r02_pmc_summary.json", "w"), indent=1)
else:
    print("no counter file")
PY
rm -rf $R/prof_pmc_fetch
echo "== other configurations"
timeout -k 10 200 python bench.py --model tinyllama --recipe gptq-4bit-128g --no-cpu-baseline --no-prefill 2>/dev/null | tail -1 > $R/r02_bench_tinyllama_gptq.json; cut -c1-200 $R/r02_bench_tinyllama_gptq.json
timeout -k 10 200 python bench.py --batch 16 --steps 32 --warmup 4 --no-cpu-baseline --no-prefill 2>/dev/null | tail -1 > $R/r02_bench_b16.json; cut -c1-200 $R/r02_bench_b16.json
timeout -k 10 300 python bench.py --model llama2-70b --recipe 2.5bpw --cache q4 --steps 32 --warmup 4 --no-cpu-baseline --no-prefill 2>/dev/null | tail -1 > $R/r02_bench_70b_q4.json; cut -c1-200 $R/r02_bench_70b_q4.json
timeout -k 10 300 python bench.py --model mixtral-8x7b --recipe 3.5bpw --steps 32 --warmup 4 --no-cpu-baseline --no-prefill 2>/dev/null | tail -1 > $R/r02_bench_mixtral_b1.json; cut -c1-200 $R/r02_bench_mixtral_b1.json
timeout -k 10 300 python bench.py --model mixtral-8x7b --recipe 3.5bpw --batch 16 --steps 32 --warmup 4 --no-cpu-baseline --no-prefill 2>/dev/null | tail -1 > $R/r02_bench_mixtral_b16.json; cut -c1-200 $R/r02_bench_mixtral_b16.json
timeout -k 10 200 python tools/moe_bench.py 2>/dev/null > $R/r02_moe_bench.jsonl; cut -c1-130 $R/r02_moe_bench.jsonl
if [ -f exllamav2_amd/libexl2_hip_trace.so ]; then timeout -k 10 200 python tools/trace_flat.py 2>/dev/null > $R/r02_trace_flat.txt; grep "waves\|span" $R/r02_trace_flat.txt; timeout -k 10 200 python tools/trace_overlap.py > $R/r02_trace_overlap.txt 2>/dev/null; grep "producer last end" $R/r02_trace_overlap.txt; fi
timeout -k 10 120 python tools/attn_prefill_bench.py 2>/dev/null > $R/r02_attn_prefill_bench.jsonl; cut -c1-160 $R/r02_attn_prefill_bench.jsonl
echo "== probes"; timeout -k 5 60 tools/probes/fork_probe 800 > $R/r02_fork_probe.txt 2>&1; tail -3 $R/r02_fork_probe.txt
(for v in off on; do echo "== kernarg preload $v"; timeout -k 5 60 tools/probes/preload_${v}_probe; done) > $R/r02_preload_probe.txt 2>&1
echo "== overlapped chain vs serial chain (same box)"
(echo "EXL2_CHAIN_OVERLAP=1 (two graphs side by side) vs 0 (serial chain), same box, bench.py --no-prefill --no-cpu-baseline --steps 128 --warmup 16"
 for ov in 1 0; do echo "== overlap=$ov"; EXL2_CHAIN_OVERLAP=$ov timeout -k 10 240 python bench.py --no-prefill --no-cpu-baseline --steps 128 --warmup 16 2>/dev/null | tail -1; done) > $R/r02_overlap_bench.txt
cut -c1-300 $R/r02_overlap_bench.txt
rocm-smi --showproductname 2>/dev/null | grep -i "card series\|gfx" | head -3 > $R/r02_gpu.txt
