#!/bin/bash
# Round 4, call A: parity of the pipelined lean kernel on the MI355X, same-box A/B of its forms, in-kernel timeline, evidence
# for today's 5-16-row route (rocprof stats + PMC), the unmodified reference host's decode rate on the drop-in.
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT/gpurun_out; cd $GRAFT_REPO_ROOT
echo "== pytest gpu"; timeout -k 10 900 python -m pytest tests -m gpu -q -x --timeout 600 > $R/r04a_pytest_gpu.log 2>&1; echo "rc=$?"; tail -4 $R/r04a_pytest_gpu.log
B="--steps 64 --warmup 8 --no-cpu-baseline --no-prefill --no-ctx-window --no-parity-check"
for rep in 1 2; do
  for v in "A=1" "EXL2_LEAN_PIPE=0" "EXL2_HIP_LIB=$GRAFT_REPO_ROOT/exllamav2_amd/libexl2_hip_pipe1.so" "EXL2_HIP_LIB=$GRAFT_REPO_ROOT/exllamav2_amd/libexl2_hip_r3form.so"; do
    echo -n "$(echo $v | sed 's#.*/##') : "; env $v timeout -k 10 200 python bench.py $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], 'tok/s', d['ms_per_step'], 'ms', d['roofline']['avg_launch_us'], 'us/launch frac', d['roofline']['frac'])"
  done
done 2>&1 | tee $R/r04a_pipe_ab.txt
echo "== full bench line with parity (default lib)"; timeout -k 10 400 python bench.py --no-cpu-baseline > $R/r04a_bench.json 2> $R/r04a_bench.err; echo "rc=$?"; cut -c1-700 $R/r04a_bench.json; tail -2 $R/r04a_bench.err
if [ -f exllamav2_amd/libexl2_hip_trace.so ]; then timeout -k 10 200 python tools/trace_lean.py > $R/r04a_trace_lean.txt 2>&1; grep -c . $R/r04a_trace_lean.txt; grep "span" $R/r04a_trace_lean.txt | head; fi
echo "== rows 5-16: today's route (bs=16), rocprof stats + PMC"
(cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/prof_b16 -o b16 -- python $GRAFT_REPO_ROOT/bench.py --batch 16 --steps 32 --warmup 4 --no-cpu-baseline --no-prefill --no-parity-check --no-ctx-window > $R/r04a_b16.log 2>&1); tail -1 $R/r04a_b16.log | cut -c1-200
head -12 $R/prof_b16/b16_kernel_stats.csv | cut -c1-150; cp $R/prof_b16/b16_kernel_stats.csv $R/r04a_b16_kernel_stats.csv; rm -rf $R/prof_b16
(cd /tmp && timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/prof_b16p -o b16 -- python $GRAFT_REPO_ROOT/bench.py --batch 16 --steps 8 --warmup 2 --no-graph --no-cpu-baseline --no-prefill --no-parity-check --no-ctx-window > $R/r04a_b16_pmc.log 2>&1); echo "rc=$?"
python - <<'PY'
import csv, glob, collections, json
agg = collections.defaultdict(list)
for f in glob.glob("gpurun_out/prof_b16p/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[(r["Kernel_Name"][:70], r["Counter_Name"])].append(float(r["Counter_Value"]))
out = {f"{c}:{k}": {"launches": len(v), "avg": sum(v) / len(v)} for (k, c), v in agg.items()}
json.dump(out, open("gpurun_out/r04a_b16_pmc_summary.json", "w"), indent=1)
by = collections.defaultdict(dict)
for (k, c), v in agg.items(): by[k][c] = (sum(v) / len(v), len(v))
for k, d in sorted(by.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CU_CYCLES", (0, 0))[0] * kv[1].get("SQ_BUSY_CU_CYCLES", (0, 0))[1])[:8]:
    print(k[:60], {c: round(a) for c, (a, n) in d.items()}, "launches", max(n for a, n in d.values()))
PY
rm -rf $R/prof_b16p
echo "== the unmodified reference host on the drop-in: 7B decode loop"
timeout -k 10 420 python tools/dropin_decode_bench.py --tokens 128 > $R/r04a_dropin_decode.json 2> $R/r04a_dropin_decode.err; echo "rc=$?"; tail -1 $R/r04a_dropin_decode.json; tail -3 $R/r04a_dropin_decode.err
rocm-smi --showproductname 2>/dev/null | grep -i "card series\|gfx" | head -3 > $R/r04a_gpu.txt
