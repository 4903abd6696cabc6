"""Timing experiments on qgemm_skinny_kernel (SK_DBG build: EXL2_HIP_LIB=exllamav2_amd/libexl2_hip_skdbg.so): one 7B linear at M rows,
every (K splits, parts left out) configuration REPS times in a row; run under rocprofv3 --kernel-trace, tools/debug/skinny_dbg_parse.py
cuts the trace into the configurations by order."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from exllamav2_amd.ext import ext_c as ext, none_tensor
from exllamav2_amd.synth import synth_linear

k, n, m = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
reps = 12
gen = torch.Generator(device="cuda"); gen.manual_seed(0)
w = synth_linear(k, n, ([4], [1.0], 128), "cuda", gen)
h = ext.make_q_matrix_from_dict(w, none_tensor)
a = torch.randn((m, k), device="cuda", dtype=torch.float16)
c = torch.empty((m, n), device="cuda", dtype=torch.float16)
for _ in range(300): ext.gemm_half_q_half(a, h, c)          # clock ramp
torch.cuda.synchronize()
configs = [(ks, dbg) for ks in (2, 4, 8) for dbg in (0, 1, 2, 3, 4, 8, 16, 31)]
for ks, dbg in configs:
    os.environ["EXL2_SKINNY_SPLITK"] = str(ks); os.environ["EXL2_SKINNY_DBG"] = str(dbg)
    for _ in range(reps): ext.gemm_half_q_half(a, h, c)
    torch.cuda.synchronize()
print(json.dumps({"configs": configs, "reps": reps, "warm": 300}))
