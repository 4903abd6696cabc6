#!/bin/bash
# Round 4, call B: the pipelined lean kernel with the buffer-form staging copies and the requests of the last items issued right
# behind them: parity, same-box A/B against the round-3 form, in-kernel timeline, the full bench line.
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT/gpurun_out; cd $GRAFT_REPO_ROOT
echo "== pytest gpu"; timeout -k 10 900 python -m pytest tests -m gpu -q -x --timeout 600 > $R/r04b_pytest_gpu.log 2>&1; echo "rc=$?"; tail -4 $R/r04b_pytest_gpu.log
B="--steps 64 --warmup 8 --no-cpu-baseline --no-prefill --no-ctx-window --no-parity-check"
for rep in 1 2; do
  for v in "A=1" "EXL2_LEAN_PIPE=0" "EXL2_HIP_LIB=$GRAFT_REPO_ROOT/exllamav2_amd/libexl2_hip_r3form.so"; do
    echo -n "$(echo $v | sed 's#.*/##') : "; env $v timeout -k 10 200 python bench.py $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], 'tok/s', d['ms_per_step'], 'ms', d['roofline']['avg_launch_us'], 'us/launch frac', d['roofline']['frac'])"
  done
done 2>&1 | tee $R/r04b_pipe_ab.txt
for b in 2 4; do echo -n "batch $b: "; timeout -k 10 200 python bench.py --batch $b --steps 32 --warmup 4 --no-cpu-baseline --no-prefill --no-parity-check --no-ctx-window 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], 'tok/s', d['ms_per_step'], 'ms')"; done 2>&1 | tee -a $R/r04b_pipe_ab.txt
echo "== full bench line with parity (default lib)"; timeout -k 10 500 python bench.py > $R/r04b_bench.json 2> $R/r04b_bench.err; echo "rc=$?"; cut -c1-400 $R/r04b_bench.json; tail -2 $R/r04b_bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04b_bench.json").read().strip().splitlines()[-1])
print("roofline", {k: d["roofline"][k] for k in ("achieved", "frac", "avg_launch_us")}); print("parity", d.get("parity_check")); print("prefill", json.dumps(d.get("prefill"))[:900]); print("cpu", json.dumps(d.get("cpu_baseline"))[:200])
PY
if [ -f exllamav2_amd/libexl2_hip_trace.so ]; then timeout -k 10 200 python tools/trace_lean.py > $R/r04b_trace_lean.txt 2>&1; grep "span\|requested\|landed\|decoded" $R/r04b_trace_lean.txt | head -30; fi
echo "== 70B 2.5bpw + Q4 cache (configs[3]) with the parity check through the Q4 cache"
timeout -k 10 400 python bench.py --model llama2-70b --recipe 2.5bpw --cache q4 --steps 32 --warmup 4 --no-cpu-baseline --no-prefill 2>$R/r04b_70b.err | tail -1 > $R/r04b_bench_70b_q4.json; cut -c1-300 $R/r04b_bench_70b_q4.json; tail -2 $R/r04b_70b.err
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r04b_bench_70b_q4.json").read().strip().splitlines()[-1]); print("70b", d["value"], d["roofline"]["frac"], d.get("parity_check"))
except Exception as e: print("70b parse", e)
PY
