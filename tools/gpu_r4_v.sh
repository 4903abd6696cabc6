#!/bin/bash
# what bounds the 16-row launches now: PMC pass (eager launches of the same kernels) over a 16-sequence decode
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $R; cd $GRAFT_REPO_ROOT
(cd /tmp && timeout -k 10 280 rocprofv3 --pmc FETCH_SIZE SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/prof_b16p -o b16 -- python $GRAFT_REPO_ROOT/bench.py --batch 16 --steps 8 --warmup 2 --headline-only --no-graph --no-parity-check > $R/r04_b16_pmc.log 2>&1); echo "rc=$?"
python - <<'PY'
import csv, glob, collections, json
agg = collections.defaultdict(list)
for f in glob.glob("gpurun_out/prof_b16p/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[(r["Kernel_Name"][:80], r["Counter_Name"])].append(float(r["Counter_Value"]))
out = {f"{c}:{k}": {"launches": len(v), "avg": sum(v) / len(v)} for (k, c), v in agg.items() if "lean" in k or "attn" in k}
json.dump(out, open("gpurun_out/r04_b16_pmc_summary.json", "w"), indent=1)
by = collections.defaultdict(dict)
for (k, c), v in agg.items(): by[k][c] = (sum(v) / len(v), len(v))
for k, d in sorted(by.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CU_CYCLES", (0, 0))[0] * kv[1].get("SQ_BUSY_CU_CYCLES", (0, 0))[1])[:6]:
    print(k[:76], {c: round(a) for c, (a, n) in d.items()}, "launches", max(n for a, n in d.values()))
PY
rm -rf $R/prof_b16p
