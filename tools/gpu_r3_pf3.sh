#!/bin/bash
# Round 3: from how many rows does decoding the weights once per call (wfrag_kernel) pay?
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT/gpurun_out; cd $GRAFT_REPO_ROOT
python - <<'PY' | tee gpurun_out/r03_prefill_wpre_threshold.txt
import os, sys, json, torch
sys.path.insert(0, os.getcwd())
from exllamav2_amd.ext import ext_c as ext, none_tensor
from exllamav2_amd.synth import synth_linear
def t(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for (k, n, rec) in ((4096, 11008, ([4], [1.0], 128)), (4096, 4096, ([5, 4], [0.1, 0.9], 128)), (11008, 4096, ([8, 4], [0.05, 0.95], [32, 128]))):
    gen = torch.Generator(device="cuda"); gen.manual_seed(0)
    w = synth_linear(k, n, rec, "cuda", gen)
    h = ext.make_q_matrix_from_dict(w, none_tensor)
    for m in (192, 256, 384, 512, 768, 1024, 1536, 2048):
        a = torch.randn((m, k), device="cuda", dtype=torch.float16); c = torch.empty((m, n), device="cuda", dtype=torch.float16)
        res = {}
        for name, env in (("in_gemm", "0"), ("once_per_call", "1")):
            os.environ["EXL2_PREFILL_WPRE_MIN_ROWS"] = env
            res[name] = round(t(lambda: ext.gemm_half_q_half(a, h, c)) * 1e3, 1)
        print(k, n, m, res, "us; speedup", round(res["in_gemm"] / res["once_per_call"], 2), flush=True)
    ext.free_q_matrix(h)
PY
