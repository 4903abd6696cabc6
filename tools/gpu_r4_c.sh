#!/bin/bash
# Round 4, call C: 70B chain route after the staging fix, the drop-in's flash_attn_func shim (parity + decode rate + host profile)
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT/gpurun_out; cd $GRAFT_REPO_ROOT
echo "== pytest (drop-in + chain)"; timeout -k 10 600 python -m pytest tests/test_dropin_reference.py tests/test_chain.py -m gpu -q -x --timeout 600 2>&1 | tail -4
echo "== 70B 2.5bpw + Q4 cache"
timeout -k 10 400 python bench.py --model llama2-70b --recipe 2.5bpw --cache q4 --steps 32 --warmup 4 --no-cpu-baseline --no-prefill 2>$R/r04c_70b.err | tail -1 > $R/r04c_bench_70b_q4.json; tail -2 $R/r04c_70b.err
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r04c_bench_70b_q4.json").read().strip().splitlines()[-1]); print("70b", d["value"], d["roofline"]["frac"], d["roofline"]["avg_launch_us"], d.get("parity_check", {}).get("decode_route"), d.get("extra"))
except Exception as e: print("70b parse", e)
PY
echo "== drop-in decode loop: flash shim, torch fallback, profile"
timeout -k 10 300 python tools/dropin_decode_bench.py --tokens 128 --attn flash > $R/r04c_dropin_flash.json 2> $R/r04c_dropin_flash.err; tail -1 $R/r04c_dropin_flash.json | cut -c1-400; tail -2 $R/r04c_dropin_flash.err | grep -v amdgpu
timeout -k 10 300 python tools/dropin_decode_bench.py --tokens 128 --attn torch > $R/r04c_dropin_torch.json 2>/dev/null; tail -1 $R/r04c_dropin_torch.json | cut -c1-300
timeout -k 10 300 python tools/dropin_decode_bench.py --tokens 64 --attn flash --profile > /dev/null 2> $R/r04c_dropin_profile.txt; grep -v amdgpu $R/r04c_dropin_profile.txt | head -45
