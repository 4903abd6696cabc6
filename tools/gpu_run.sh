#!/bin/bash
# ONE parameterised runner for every GPU call (replaces the per-call gpu_r*.sh scripts of rounds 1-4; their history is in git).
#   gpurun --timeout 900 -- 'TAG=r05 STAGES="smoke tests bench stats" tools/gpu_run.sh'
# Outputs go to gpurun_out/ (scratch); what is judged is copied to profiles/ by hand afterwards.
#
# STAGES (any subset, in this order):
#   smoke      __graft_entry__.smoke()
#   tests      pytest -m gpu            (TESTS="tests/test_x.py -k expr" narrows it; default: all of tests/)
#   bench      bench.py with default flags + with the driver's flags (--gpus 1 --steps 20 --warmup 5); bench_default / bench_driver: one of them
#   stats      rocprofv3 --kernel-trace --stats of the headline decode loop  -> ${TAG}_kernel_stats.csv
#   pmc        rocprofv3 --pmc FETCH_SIZE (own pass, kernel-trace only) over the headline loop, graph replay first and eager
#              launches if that does not finish -> ${TAG}_pmc_summary.json
#   configs    the other BASELINE configurations as bench lines (tinyllama gptq, batch 4/8/16, q4 cache, 70B, mixtral b1/b16)
#   micro      tools/moe_bench.py, prefill_bench.py, attn_prefill_bench.py, sampler_bench.py
#   trace      in-kernel timeline of the chained decode kernel (needs tools/build_trace.sh's library)
#   ab         same-box A/B: VARIANTS="name=ENV1=v,ENV2=v ..." (name `head` = no env); libraries built by tools/build_variant.sh are
#              selected with EXL2_HIP_LIB=...; REPS (default 2) rounds, interleaved; AB_FLAGS = extra bench flags
#   probes     every built binary under tools/probes/*_probe (PROBES="a_probe b_probe" narrows it)
#   pfpmc      rocprofv3 --pmc over tools/prefill_bench.py (MFMA busy, LDS, wait states: two passes) -> ${TAG}_pmc_prefill_summary.json
mkdir -p gpurun_out; export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; R=$ROOT/gpurun_out; T=${TAG:-r05}
STAGES=${STAGES:-"smoke tests bench"}
cd $ROOT
has() { case " $STAGES " in *" $1 "*) return 0;; esac; return 1; }
line() { timeout -k 10 ${2:-300} python bench.py $1 2>$R/${T}_last.err | tail -1; }

if has smoke; then echo "== smoke"; timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $R/${T}_smoke.log 2>&1; echo "rc=$?"; tail -1 $R/${T}_smoke.log; fi
if has tests; then
  echo "== pytest -m gpu ${TESTS:-tests}"
  timeout -k 10 ${TESTS_TIMEOUT:-1200} python -m pytest ${TESTS:-tests} -m gpu -q --timeout 600 ${TESTS_X--x} > $R/${T}_pytest_gpu.log 2>&1; echo "rc=$?"; tail -${TESTS_TAIL:-4} $R/${T}_pytest_gpu.log
fi
if has bench || has bench_default; then
  echo "== bench (default flags)"; timeout -k 10 900 python bench.py > $R/${T}_bench.json 2> $R/${T}_bench.err; echo "rc=$?"; cut -c1-1800 $R/${T}_bench.json; tail -3 $R/${T}_bench.err
fi
if has bench || has bench_driver; then
  echo "== bench (the driver's flags)"; timeout -k 10 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $R/${T}_bench_driver_flags.json 2>/dev/null; cut -c1-400 $R/${T}_bench_driver_flags.json
fi
if has stats; then
  echo "== rocprofv3 kernel stats (bs=1 headline loop)"
  (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/prof_stats -o $T -- python $ROOT/bench.py --steps 64 --warmup 8 --windows 1 --headline-only --no-parity-check > $R/${T}_rocprof_stats.log 2>&1); echo "rc=$?"
  f=$(find $R/prof_stats -name "${T}_kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $R/${T}_kernel_stats.csv && head -8 $f | cut -c1-170
  rm -rf $R/prof_stats
fi
if has pmc; then
  for mode in graph nograph; do
    flag=""; [ $mode = nograph ] && flag="--no-graph"
    echo "== rocprofv3 --pmc FETCH_SIZE ($mode)"
    (cd /tmp && EXL2_BENCH_RAMP_S=0 timeout -k 10 ${PMC_TIMEOUT:-240} rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/prof_pmc -o $T -- python $ROOT/bench.py --steps ${PMC_STEPS:-4} --warmup 1 --windows 1 --headline-only $flag --no-parity-check > $R/${T}_rocprof_pmc_$mode.log 2>&1); rc=$?; echo "rc=$rc"
    python - $mode $T <<'PY'
import csv, glob, collections, json, sys
mode, tag = sys.argv[1], sys.argv[2]
out = {}
fs = glob.glob("gpurun_out/prof_pmc/**/*counter_collection.csv", recursive=True)
if fs:
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        agg[(r["Kernel_Name"][:64], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:8]:
        print(k, c, "launches", len(v), "avg", round(sum(v) / len(v), 1))
        out[f"{c}:{k}"] = {"launches": len(v), "avg": sum(v) / len(v), "state": mode}
    if any("qgemv_lean" in k for k in out):
        json.dump(out, open(f"gpurun_out/{tag}_pmc_summary_{mode}.json", "w"), indent=1)
else:
    print("no counter file")
PY
    rm -rf $R/prof_pmc
    [ -f $R/${T}_pmc_summary_graph.json ] && break        # the timed configuration itself was profiled: no need for the eager pass
  done
fi
if has configs; then
  echo "== other configurations"
  for c in "tinyllama_gptq|--model tinyllama --recipe gptq-4bit-128g --headline-only" \
           "b4|--batch 4 --steps 32 --warmup 4 --headline-only" "b8|--batch 8 --steps 32 --warmup 4 --headline-only" \
           "b16|--batch 16 --steps 32 --warmup 4 --headline-only" "7b_q4cache|--cache q4 --steps 64 --warmup 8 --headline-only" \
           "70b_q4|--model llama2-70b --recipe 2.5bpw --cache q4 --steps 32 --warmup 4 --headline-only" \
           "mixtral_b1|--model mixtral-8x7b --recipe 3.5bpw --steps 32 --warmup 4 --headline-only" \
           "mixtral_b16|--model mixtral-8x7b --recipe 3.5bpw --batch 16 --steps 32 --warmup 4 --headline-only"; do
    name=${c%%|*}; flags=${c#*|}
    case " ${CONFIGS:-all} " in *" all "*|*" $name "*) ;; *) continue;; esac
    line "$flags" 600 > $R/${T}_bench_$name.json; echo "$name: $(cut -c1-260 $R/${T}_bench_$name.json)"; [ -s $R/${T}_bench_$name.json ] || tail -2 $R/${T}_last.err
  done
fi
if has micro; then
  timeout -k 10 200 python tools/moe_bench.py 2>/dev/null > $R/${T}_moe_bench.jsonl; cut -c1-130 $R/${T}_moe_bench.jsonl
  timeout -k 10 400 python tools/prefill_bench.py 2>/dev/null > $R/${T}_prefill_gemm.jsonl; cut -c1-200 $R/${T}_prefill_gemm.jsonl | tail -12
  timeout -k 10 200 python tools/attn_prefill_bench.py 2>/dev/null > $R/${T}_attn_prefill_bench.jsonl; cut -c1-160 $R/${T}_attn_prefill_bench.jsonl | tail -4
  timeout -k 10 200 python tools/sampler_bench.py 2>/dev/null > $R/${T}_sampler_bench.jsonl; cut -c1-160 $R/${T}_sampler_bench.jsonl | tail -4
fi
if has trace && [ -f exllamav2_amd/libexl2_hip_trace.so ]; then
  timeout -k 10 300 python tools/trace_lean.py > $R/${T}_trace_lean.txt 2>&1; grep -i "waves\|span\|launch" $R/${T}_trace_lean.txt | head -40
fi
if has ab; then
  echo "== same-box A/B ($VARIANTS)"
  for rep in $(seq 1 ${REPS:-2}); do
    for v in ${VARIANTS:-head}; do
      name=${v%%=*}; envs=""; [ "$name" != "$v" ] && envs=$(echo "${v#*=}" | tr ',' ' ')
      echo -n "$name : "; env $envs timeout -k 10 300 python bench.py --steps ${AB_STEPS:-256} --warmup 16 --headline-only --no-parity-check $AB_FLAGS 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d.get('roofline', {})
print(d['value'], 'tok/s', d.get('windows', {}).get('tokens_per_s'), r.get('avg_launch_us'), 'us/launch', r.get('frac'))"
    done
  done 2>&1 | tee $R/${T}_ab_${AB_NAME:-variants}.txt
fi
if has pfpmc; then
  echo "== rocprofv3 --pmc over the prefill GEMM (two passes)"
  (cd /tmp && timeout -k 10 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/prof_pmc_pf -o a -- python $ROOT/tools/prefill_bench.py --quick --variants auto --reps 2 > $R/${T}_pmc_pf_a.log 2>&1); echo "rc=$?"
  (cd /tmp && timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d $R/prof_pmc_pf2 -o b -- python $ROOT/tools/prefill_bench.py --quick --variants auto --reps 2 > $R/${T}_pmc_pf_b.log 2>&1); echo "rc=$?"
  python - $T <<'PY'
import csv, glob, collections, json, sys
tag = sys.argv[1]
out = {}
for d in ("prof_pmc_pf", "prof_pmc_pf2"):
    for f in glob.glob(f"gpurun_out/{d}/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "qgemm_mfma" in r["Kernel_Name"] or "wfrag" in r["Kernel_Name"] or "qgemm_prefill" in r["Kernel_Name"]:
                agg[(r["Kernel_Name"][:60], r["Counter_Name"], r.get("Grid_Size", ""))].append(float(r["Counter_Value"]))
        for (k, c, g), v in sorted(agg.items()):
            out[f"{c}:{k}:grid{g}"] = {"launches": len(v), "avg": sum(v) / len(v)}
json.dump(out, open(f"gpurun_out/{tag}_pmc_prefill_summary.json", "w"), indent=1)
by = collections.defaultdict(dict)
for key, v in out.items():
    c, k, g = key.split(":", 2)[0], key.split(":", 2)[1], key.rsplit(":", 1)[1]
    by[(k, g)][c] = v["avg"]
lines = []
for (k, g), d in by.items():
    if "SQ_BUSY_CU_CYCLES" in d:
        line = f"{k} {g}: "
        if "SQ_VALU_MFMA_BUSY_CYCLES" in d: line += "MFMA busy %.1f%%  " % (100 * d["SQ_VALU_MFMA_BUSY_CYCLES"] / (4 * d["SQ_BUSY_CU_CYCLES"]))
        if "SQ_LDS_IDX_ACTIVE" in d: line += "LDS idx active / CU busy %.2f  bank conflict cycles / active %.3f  " % (d["SQ_LDS_IDX_ACTIVE"] / d["SQ_BUSY_CU_CYCLES"], d.get("SQ_LDS_BANK_CONFLICT", 0) / max(1, d["SQ_LDS_IDX_ACTIVE"]))
        if "SQ_WAIT_ANY" in d: line += "wave wait %.0f%% (inst-any wait %.0f%%, LDS-inst wait %.0f%%) active LDS %.0f%% VALU %.0f%%" % (100 * d["SQ_WAIT_ANY"] / d["SQ_WAVE_CYCLES"], 100 * d["SQ_WAIT_INST_ANY"] / d["SQ_WAVE_CYCLES"], 100 * d["SQ_WAIT_INST_LDS"] / d["SQ_WAVE_CYCLES"], 100 * d["SQ_ACTIVE_INST_LDS"] / d["SQ_WAVE_CYCLES"], 100 * d["SQ_ACTIVE_INST_VALU"] / d["SQ_WAVE_CYCLES"])
        lines.append(line)
open(f"gpurun_out/{tag}_pmc_prefill_lines.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
  rm -rf $R/prof_pmc_pf $R/prof_pmc_pf2
fi
if has probes; then
  plist="tools/probes/*_probe"; [ -n "$PROBES" ] && plist=$(for q in $PROBES; do echo tools/probes/$q; done)
  for p in $plist; do [ -x $p ] && { echo "== $p"; timeout -k 10 120 $p 2>&1 | tee $R/${T}_$(basename $p).txt | tail -${PROBE_TAIL:-12}; }; done
fi
rocm-smi --showproductname 2>/dev/null | grep -i "card series\|gfx" | head -3 > $R/${T}_gpu.txt
echo "== done"
if has skinny; then
  echo "== 17-128-row q_gemm by kernel time (tools/skinny_bench.py)"
  (cd /tmp && timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d $R/prof_skb -o skb -- python $ROOT/tools/skinny_bench.py > $R/${T}_skinny_meta.json 2>$R/${T}_skinny.err); echo "rc=$?"
  f=$(find $R/prof_skb -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python tools/skinny_bench.py --parse $f $R/${T}_skinny_meta.json | tee $R/${T}_skinny_bench.jsonl
  rm -rf $R/prof_skb
fi
