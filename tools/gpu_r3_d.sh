#!/bin/bash
# Round 3, GPU call D: where do the lean kernel's waves spend their cycles?  SQ counters of the decode loop (two PMC passes)
# next to the same counters of the skeleton probe, plus the decode line.
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
rocprofv3 -L 2>/dev/null | grep -o "SQ[C]*_[A-Z0-9_]*" | sort -u | tr '\n' ' ' > $R/r03d_counters.txt; wc -w $R/r03d_counters.txt
B="python $GRAFT_REPO_ROOT/bench.py --steps 32 --warmup 4 --no-cpu-baseline --no-prefill --no-ctx-window --no-parity-check --no-graph"
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU"
P2="SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_IFETCH SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS"
i=0
for P in "$P1" "$P2"; do i=$((i+1))
  (cd /tmp && EXL2_LEAN_OCC=${OCC:-6} timeout -k 10 200 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $R/pmc_lean_$i -o p -- $B > $R/r03d_pmc_lean_$i.log 2>&1); echo "lean pass $i rc=$?"
  (cd /tmp && timeout -k 10 200 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $R/pmc_probe_$i -o p -- $GRAFT_REPO_ROOT/tools/probes/lean_probe > $R/r03d_pmc_probe_$i.log 2>&1); echo "probe pass $i rc=$?"
done
python - <<'PY'
import csv, glob, collections, json
out = {}
for tag in ("lean", "probe"):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for i in (1, 2):
        for f in glob.glob(f"gpurun_out/pmc_{tag}_{i}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"]
                if "qgemv_lean" in k or "lean_kernel" in k:
                    key = (k[:70], r["Counter_Name"])
                    agg[key][0] += 1; agg[key][1] += float(r["Counter_Value"])
    for (k, c), (n, v) in sorted(agg.items()):
        out.setdefault(tag, {}).setdefault(k, {})[c] = round(v / n, 1)
        out[tag][k]["launches"] = n
json.dump(out, open("gpurun_out/r03d_pmc_summary.json", "w"), indent=1)
for tag in out:
    for k, d in out[tag].items():
        print(tag, k, json.dumps(d))
PY
rm -rf $R/pmc_lean_* $R/pmc_probe_*
echo "== bench"; for v in "EXL2_LEAN_OCC=6" "EXL2_LEAN_OCC=8"; do env $v timeout -k 10 200 python bench.py --steps 64 --warmup 8 --no-cpu-baseline --no-prefill --no-ctx-window --no-parity-check > $R/r03d_bench_$v.json 2>/dev/null; python - "$R/r03d_bench_$v.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1][-20:], d["value"], "tok/s", "avg launch us", d["roofline"]["avg_launch_us"], "frac", d["roofline"]["frac"])
except Exception as e: print("no json", e)
PY
done
