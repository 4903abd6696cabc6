#!/bin/bash
# Round 3, GPU call E: same-box A/B of library variants (tools/build_variant.sh) x register budgets, + I-cache counters.
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
B="python bench.py --steps 64 --warmup 8 --no-cpu-baseline --no-prefill --no-ctx-window --no-parity-check"
for lib in ${LIBS:-default normreg normreg_noprescale noprescale nolow normreg_nolow}; do
  for occ in ${OCCS:-6 8}; do
    L=$GRAFT_REPO_ROOT/exllamav2_amd/libexl2_hip_$lib.so; [ "$lib" = default ] && L=$GRAFT_REPO_ROOT/exllamav2_amd/libexl2_hip.so
    EXL2_HIP_LIB=$L EXL2_LEAN_OCC=$occ timeout -k 10 200 $B > $R/r03e_bench_${lib}_$occ.json 2>/dev/null; python - "$R/r03e_bench_${lib}_$occ.json" "$lib occ$occ" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print("%-28s" % sys.argv[2], d["value"], "tok/s", "avg launch us", d["roofline"]["avg_launch_us"], "frac", d["roofline"]["frac"])
except Exception as e: print(sys.argv[2], "no json", e)
PY
  done
done
P="SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVES SQ_BUSY_CYCLES"
(cd /tmp && EXL2_LEAN_OCC=6 timeout -k 10 200 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $R/pmc_ic -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 32 --warmup 4 --no-cpu-baseline --no-prefill --no-ctx-window --no-parity-check --no-graph > $R/r03e_pmc_ic.log 2>&1); echo "pmc rc=$?"
python - <<'PY'
import csv, glob, collections, json
agg = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob("gpurun_out/pmc_ic/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "qgemv_lean" in k or "attn_fused" in k:
            agg[(k[:60], r["Counter_Name"])][0] += 1; agg[(k[:60], r["Counter_Name"])][1] += float(r["Counter_Value"])
out = {}
for (k, c), (n, v) in sorted(agg.items()): out.setdefault(k, {})[c] = round(v / n, 1)
for k, d in out.items(): print(k, json.dumps(d))
json.dump(out, open("gpurun_out/r03e_pmc_icache.json", "w"), indent=1)
PY
rm -rf $R/pmc_ic
