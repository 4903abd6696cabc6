#!/bin/bash
# First GPU call of round 3: re-establish the evidence of the round-2 source state on one MI355X in one go --
# smoke, all -m gpu parity tests, the default bench line (decode + prefill + both CPU-baseline variants in their child process),
# rocprofv3 kernel stats of the decode loop and of the prefill tool, the PMC FETCH_SIZE pass, sampler timing.
# Outputs -> gpurun_out/ (copy what is to be judged into profiles/).  ~4-5 GPU-minutes.
# (probe binaries: run tools/probes/build.sh in the build container first)
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
echo "== smoke"; timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > $R/r03_smoke.log 2>&1; echo "rc=$?"; tail -1 $R/r03_smoke.log
echo "== pytest gpu"; timeout -k 10 300 python -m pytest tests -m gpu -q --timeout 240 > $R/r03_pytest_gpu.log 2>&1; echo "rc=$?"; tail -3 $R/r03_pytest_gpu.log
echo "== bench"; S=$(date +%s); timeout -k 10 600 python bench.py > $R/r03_bench.json 2> $R/r03_bench.err; echo "rc=$? wall=$(( $(date +%s) - S ))s"; cut -c1-600 $R/r03_bench.json
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r03_bench.json").read().strip().splitlines()[-1])
print("prefill", d.get("prefill")); print("cpu_baseline", json.dumps(d.get("cpu_baseline"))[:1200])
PY
echo "== rocprof stats (bs=1 decode)"; (cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/prof_stats -o r03 -- python $GRAFT_REPO_ROOT/bench.py --steps 64 --warmup 8 --no-cpu-baseline --no-prefill --no-parity-check --no-ctx-window > $R/r03_rocprof_stats.log 2>&1); echo "rc=$?"
head -8 $R/prof_stats/r03_kernel_stats.csv | cut -c1-160; rm -f $R/prof_stats/*kernel_trace.csv
echo "== rocprof pmc FETCH_SIZE"; (cd /tmp && timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/prof_pmc_fetch -o r03 -- python $GRAFT_REPO_ROOT/bench.py --steps 16 --warmup 2 --no-cpu-baseline --no-prefill --no-graph --no-parity-check --no-ctx-window > $R/r03_rocprof_pmc.log 2>&1); echo "rc=$?"
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
    json.dump(out, open("gpurun_out/r03_pmc_summary.json", "w"), indent=1)
else:
    print("no counter file")
PY
rm -rf $R/prof_pmc_fetch
echo "== sampler / prefill tools"
timeout -k 5 60 python tools/sampler_bench.py > $R/r03_sampler_bench.jsonl 2>/dev/null; cut -c1-160 $R/r03_sampler_bench.jsonl | head -4
timeout -k 10 200 python tools/prefill_bench.py > $R/r03_prefill_gemm.jsonl 2>/dev/null; cut -c1-200 $R/r03_prefill_gemm.jsonl | head -4
echo "== memory-side cache probe"; timeout -k 5 90 tools/probes/mall_probe > $R/r03_mall_probe.txt 2>&1; cat $R/r03_mall_probe.txt
echo "== persistent-kernel upper bound"; timeout -k 5 60 tools/probes/persist_probe > $R/r03_persist_probe.txt 2>&1; cat $R/r03_persist_probe.txt; timeout -k 5 120 tools/probes/chain_probe 2>&1 | head -3 >> $R/r03_persist_probe.txt
rocm-smi --showproductname 2>/dev/null | grep -i "card series\|gfx" | head -3 > $R/r03_gpu.txt
