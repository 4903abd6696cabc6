#!/bin/bash
# Round-2 closing evidence on one MI355X, ordered by importance (the GPU budget may cut the tail): all -m gpu parity tests,
# prefill q_gemm kernels side by side, the headline bench line, a kernel-argument placement A/B, rocprofv3 kernel stats of the
# bench and of the prefill GEMMs, one PMC pass on the prefill GEMM.  Outputs -> gpurun_out/ (copied to profiles/ afterwards).
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
T0=$(date +%s); lap() { echo "== [$(( $(date +%s) - T0 )) s] $1"; }
lap "pytest gpu"; timeout -k 10 420 python -m pytest tests -m gpu -q --timeout 300 > $R/r02f_pytest_gpu.log 2>&1; echo "rc=$?"; tail -5 $R/r02f_pytest_gpu.log | cut -c1-300
lap "prefill q_gemm variants"; timeout -k 10 150 python tools/prefill_bench.py --reps 3 > $R/r02f_prefill_gemm.jsonl 2> $R/r02f_prefill_gemm.err; echo "rc=$?"; cut -c1-700 $R/r02f_prefill_gemm.jsonl; tail -2 $R/r02f_prefill_gemm.err
lap "bench"; timeout -k 10 400 python bench.py > $R/r02f_bench.json 2> $R/r02f_bench.err; echo "rc=$?"; cut -c1-3000 $R/r02f_bench.json; tail -3 $R/r02f_bench.err
lap "kernarg placement A/B"
(for v in 0 1; do echo "== HIP_FORCE_DEV_KERNARG=$v"; HIP_FORCE_DEV_KERNARG=$v timeout -k 10 120 python bench.py --no-prefill --no-cpu-baseline --no-parity-check --no-ctx-window --steps 128 --warmup 16 2>/dev/null | tail -1 | cut -c1-400; done) > $R/r02f_kernarg_ab.txt; cat $R/r02f_kernarg_ab.txt
lap "rocprof stats (prefill GEMMs)"; (cd /tmp && timeout -k 10 150 rocprofv3 --kernel-trace --stats --output-format csv -d $R/prof_pf -o r02f_prefill -- python $GRAFT_REPO_ROOT/tools/prefill_bench.py --quick --variants auto --reps 3 > $R/r02f_rocprof_prefill.log 2>&1); echo "rc=$?"
head -6 $R/prof_pf/r02f_prefill_kernel_stats.csv | cut -c1-200; rm -f $R/prof_pf/*kernel_trace.csv
lap "rocprof stats (bench, bs=1)"; (cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/prof_stats -o r02f -- python $GRAFT_REPO_ROOT/bench.py --steps 64 --warmup 8 --no-cpu-baseline --no-prefill --no-parity-check --no-ctx-window > $R/r02f_rocprof_stats.log 2>&1); echo "rc=$?"
head -6 $R/prof_stats/r02f_kernel_stats.csv | cut -c1-200; rm -f $R/prof_stats/*kernel_trace.csv
lap "rocprof pmc (prefill GEMM: MFMA busy, LDS)"; (cd /tmp && timeout -k 10 150 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/prof_pmc_pf -o r02f -- python $GRAFT_REPO_ROOT/tools/prefill_bench.py --quick --variants auto --reps 2 > $R/r02f_rocprof_pmc_prefill.log 2>&1); echo "rc=$?"
python - <<'PY'
import csv, glob, collections, json
out = {}
fs = glob.glob("gpurun_out/prof_pmc_pf/**/*counter_collection.csv", recursive=True)
if fs:
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        agg[(r["Kernel_Name"][:48], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in sorted(agg.items()):
        if "qgemm" in k or "stage_rows" in k:
            print(k, c, "launches", len(v), "avg", round(sum(v) / len(v), 1))
            out[f"{c}:{k}"] = {"launches": len(v), "avg": sum(v) / len(v)}
    json.dump(out, open("gpurun_out/r02f_pmc_prefill_summary.json", "w"), indent=1)
else:
    print("no counter file");
PY
tail -3 $R/r02f_rocprof_pmc_prefill.log | cut -c1-300
rm -rf $R/prof_pmc_pf
lap "rocprof pmc FETCH_SIZE (bench)"; (cd /tmp && timeout -k 10 150 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/prof_pmc_fetch -o r02f -- python $GRAFT_REPO_ROOT/bench.py --steps 16 --warmup 2 --no-cpu-baseline --no-prefill --no-graph --no-parity-check --no-ctx-window > $R/r02f_rocprof_pmc.log 2>&1); echo "rc=$?"
python - <<'PY'
import csv, glob, collections, json
out = {}
fs = glob.glob("gpurun_out/prof_pmc_fetch/**/*counter_collection.csv", recursive=True)
if fs:
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        agg[(r["Kernel_Name"][:64], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:6]:
        print(k, c, "launches", len(v), "avg", round(sum(v) / len(v), 1))
        out[f"{c}:{k}"] = {"launches": len(v), "avg": sum(v) / len(v)}
    json.dump(out, open("gpurun_out/r02f_pmc_summary.json", "w"), indent=1)
else:
    print("no counter file")
PY
rm -rf $R/prof_pmc_fetch
rocm-smi --showproductname 2>/dev/null | grep -i "card series\|gfx" | head -3 > $R/r02f_gpu.txt
lap "done"
