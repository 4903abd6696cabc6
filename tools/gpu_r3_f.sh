#!/bin/bash
# Round 3, GPU call F: lean v5 -- parity, decode line at the register budgets, instruction counters, timeline.
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
echo "== pytest gpu (chain, model)"; timeout -k 10 300 python -m pytest tests/test_chain.py tests/test_model.py -m gpu -q --timeout 200 -x > $R/r03f_pytest.log 2>&1; echo "rc=$?"; tail -3 $R/r03f_pytest.log
B="python bench.py --steps 64 --warmup 8 --no-cpu-baseline --no-prefill --no-ctx-window --no-parity-check"
for lib in ${LIBS:-default nolow}; do
  for occ in ${OCCS:-4 6 8}; do
    L=$GRAFT_REPO_ROOT/exllamav2_amd/libexl2_hip_$lib.so; [ "$lib" = default ] && L=$GRAFT_REPO_ROOT/exllamav2_amd/libexl2_hip.so
    EXL2_HIP_LIB=$L EXL2_LEAN_OCC=$occ timeout -k 10 200 $B > $R/r03f_bench_${lib}_$occ.json 2>/dev/null; python - "$R/r03f_bench_${lib}_$occ.json" "$lib occ$occ" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print("%-28s" % sys.argv[2], d["value"], "tok/s", "avg launch us", d["roofline"]["avg_launch_us"], "frac", d["roofline"]["frac"])
except Exception as e: print(sys.argv[2], "no json", e)
PY
  done
done
P="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS"
(cd /tmp && EXL2_LEAN_OCC=6 timeout -k 10 200 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $R/pmc_f -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 32 --warmup 4 --no-cpu-baseline --no-prefill --no-ctx-window --no-parity-check --no-graph > $R/r03f_pmc.log 2>&1); echo "pmc rc=$?"
python - <<'PY'
import csv, glob, collections, json
agg = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob("gpurun_out/pmc_f/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "qgemv_lean" in k:
            agg[(k[:60], r["Counter_Name"])][0] += 1; agg[(k[:60], r["Counter_Name"])][1] += float(r["Counter_Value"])
out = {}
for (k, c), (n, v) in sorted(agg.items()): out.setdefault(k, {})[c] = round(v / n, 1)
for k, d in out.items():
    w = d.get("SQ_WAVES", 1)
    print(k, "per wave: VALU %.0f SALU %.0f SMEM %.1f LDS %.1f wave-cycles(x4) %.0f wait %.0f%%" % (d["SQ_INSTS_VALU"] / w, d["SQ_INSTS_SALU"] / w, d["SQ_INSTS_SMEM"] / w, d["SQ_INSTS_LDS"] / w, d["SQ_WAVE_CYCLES"] / w, 100 * d["SQ_WAIT_ANY"] / d["SQ_WAVE_CYCLES"]))
json.dump(out, open("gpurun_out/r03f_pmc_sq.json", "w"), indent=1)
PY
rm -rf $R/pmc_f
echo "== trace"; EXL2_LEAN_OCC=8 timeout -k 10 200 python tools/trace_lean.py > $R/r03f_trace_lean.txt 2>&1; echo "rc=$?"; grep -v "amdgpu.ids" $R/r03f_trace_lean.txt
