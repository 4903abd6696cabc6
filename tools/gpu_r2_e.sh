#!/bin/bash
# Overlapped chain (EXL2_CHAIN_OVERLAP=1): parity first (bit-identical to the serial chain), then the headline bench + a kernel trace
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_chain.py -m gpu -x -q -k overlapped 2>&1 | tail -15 | tee gpurun_out/e_pytest.txt
echo "== EXL2_CHAIN_OVERLAP=1"
EXL2_CHAIN_OVERLAP=1 timeout 200 python bench.py --no-prefill --no-cpu-baseline --no-ctx-window --steps 64 --warmup 8 2> gpurun_out/e_bench_1.err | tee gpurun_out/e_bench_1.json | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d['value'], 'tok/s', d['ms_per_step'], 'ms/step', 'roofline', d['roofline'].get('frac'), d.get('extra'))"
tail -3 gpurun_out/e_bench_1.err
cd /tmp
EXL2_CHAIN_OVERLAP=1 timeout 250 rocprofv3 --kernel-trace -d /tmp/prof_e -o ov -- python $GRAFT_REPO_ROOT/bench.py --no-prefill --no-cpu-baseline --no-ctx-window --no-parity-check --steps 8 --warmup 2 > /tmp/prof_e.log 2>&1
f=$(find /tmp/prof_e -name "*kernel_trace.csv" | head -1)
echo "trace: $f"; wc -l $f
python - <<PY
import csv
rows=list(csv.DictReader(open("$f")))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# keep the last 400 kernel records (steady-state graph replays)
keep=rows[-400:]
t0=int(keep[0]["Start_Timestamp"])
with open("$GRAFT_REPO_ROOT/gpurun_out/e_trace_tail.txt","w") as f:
    for r in keep:
        f.write("%9.2f %9.2f q%s %s\n"%((int(r["Start_Timestamp"])-t0)/1e3,(int(r["End_Timestamp"])-t0)/1e3,r.get("Queue_Id","?"),r["Kernel_Name"][:60]))
PY
