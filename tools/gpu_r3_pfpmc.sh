#!/bin/bash
# Round 3: PMC of the decode-once prefill GEMM: MFMA busy, LDS activity / conflicts, wait states (two separate passes)
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT/gpurun_out; cd $GRAFT_REPO_ROOT
(cd /tmp && timeout -k 10 150 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/prof_pmc_pf -o a -- python $GRAFT_REPO_ROOT/tools/prefill_bench.py --quick --variants auto --reps 2 > $R/r03_pmc_pf_a.log 2>&1); echo "rc=$?"
(cd /tmp && timeout -k 10 150 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d $R/prof_pmc_pf2 -o b -- python $GRAFT_REPO_ROOT/tools/prefill_bench.py --quick --variants auto --reps 2 > $R/r03_pmc_pf_b.log 2>&1); echo "rc=$?"
python - <<'PY'
import csv, glob, collections, json
out = {}
for d in ("prof_pmc_pf", "prof_pmc_pf2"):
    for f in glob.glob(f"gpurun_out/{d}/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "qgemm_mfma" in r["Kernel_Name"] or "wfrag" in r["Kernel_Name"]:
                agg[(r["Kernel_Name"][:60], r["Counter_Name"], r.get("Grid_Size", ""))].append(float(r["Counter_Value"]))
        for (k, c, g), v in sorted(agg.items()):
            out[f"{c}:{k}:grid{g}"] = {"launches": len(v), "avg": sum(v) / len(v)}
json.dump(out, open("gpurun_out/r03_pmc_prefill_summary.json", "w"), indent=1)
by = collections.defaultdict(dict)
for key, v in out.items():
    c, k, g = key.split(":", 2)[0], key.split(":", 2)[1], key.rsplit(":", 1)[1]
    by[(k, g)][c] = v["avg"]
for (k, g), d in by.items():
    if "SQ_BUSY_CU_CYCLES" in d:
        line = f"{k} {g}: "
        if "SQ_VALU_MFMA_BUSY_CYCLES" in d: line += "MFMA busy %.1f%%  " % (100 * d["SQ_VALU_MFMA_BUSY_CYCLES"] / (4 * d["SQ_BUSY_CU_CYCLES"]))
        if "SQ_LDS_IDX_ACTIVE" in d: line += "LDS idx active / CU busy %.2f  bank conflict cycles / active %.3f  " % (d["SQ_LDS_IDX_ACTIVE"] / d["SQ_BUSY_CU_CYCLES"], d.get("SQ_LDS_BANK_CONFLICT", 0) / max(1, d["SQ_LDS_IDX_ACTIVE"]))
        if "SQ_WAIT_ANY" in d: line += "wave wait %.0f%% (inst-any wait %.0f%%, LDS-inst wait %.0f%%) active LDS %.0f%% VALU %.0f%%" % (100 * d["SQ_WAIT_ANY"] / d["SQ_WAVE_CYCLES"], 100 * d["SQ_WAIT_INST_ANY"] / d["SQ_WAVE_CYCLES"], 100 * d["SQ_WAIT_INST_LDS"] / d["SQ_WAVE_CYCLES"], 100 * d["SQ_ACTIVE_INST_LDS"] / d["SQ_WAVE_CYCLES"], 100 * d["SQ_ACTIVE_INST_VALU"] / d["SQ_WAVE_CYCLES"])
        print(line)
PY
rm -rf $R/prof_pmc_pf $R/prof_pmc_pf2
