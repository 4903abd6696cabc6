#!/bin/bash
# diagnosis of the prefill path: wave wait states of the 256-column MFMA kernel (PMC), kernel breakdown of a 4-layer 8 x 2048 prefill
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
T0=$(date +%s); lap() { echo "== [$(( $(date +%s) - T0 )) s] $1"; }
lap "pmc wait states"; (cd /tmp && timeout -k 10 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d $R/prof_pmc_w -o r02g -- python $GRAFT_REPO_ROOT/tools/prefill_bench.py --quick --variants auto --reps 2 > $R/r02g_pmc_wait.log 2>&1); echo "rc=$?"
python - <<'PY'
import csv, glob, collections, json
out = {}
fs = glob.glob("gpurun_out/prof_pmc_w/**/*counter_collection.csv", recursive=True)
if fs:
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        agg[(r["Kernel_Name"][:48], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in sorted(agg.items()):
        if "qgemm" in k:
            print(k, c, "launches", len(v), "avg", round(sum(v) / len(v), 1))
            out[f"{c}:{k}"] = {"launches": len(v), "avg": sum(v) / len(v)}
    json.dump(out, open("gpurun_out/r02g_pmc_wait_summary.json", "w"), indent=1)
else:
    print("no counter file")
PY
tail -2 $R/r02g_pmc_wait.log | cut -c1-200; rm -rf $R/prof_pmc_w
lap "rocprof stats (4-layer prefill)"; (cd /tmp && timeout -k 10 150 rocprofv3 --kernel-trace --stats --output-format csv -d $R/prof_pfm -o r02g_pfmodel -- python $GRAFT_REPO_ROOT/tools/prefill_bench.py --model --layers 4 > $R/r02g_pfmodel.log 2>&1); echo "rc=$?"
head -24 $R/prof_pfm/r02g_pfmodel_kernel_stats.csv | cut -c1-150; rm -f $R/prof_pfm/*kernel_trace.csv; tail -1 $R/r02g_pfmodel.log | cut -c1-300
lap "done"
