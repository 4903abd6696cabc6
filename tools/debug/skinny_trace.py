"""In-kernel timeline of qgemm_skinny_kernel (SK_DBG build, EXL2_HIP_LIB=exllamav2_amd/libexl2_hip_skdbg.so): 100 MHz stamps per workgroup
(0 entry, 1 group map staged, 2 K loop done, 3 partial stored, 4 ticket back, 5 reduce done (last arriver), 6 end)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from exllamav2_amd.ext import ext_c as ext, none_tensor
from exllamav2_amd.synth import synth_linear

k, n, m = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
gen = torch.Generator(device="cuda"); gen.manual_seed(0)
w = synth_linear(k, n, ([4], [1.0], 128), "cuda", gen)
h = ext.make_q_matrix_from_dict(w, none_tensor)
a = torch.randn((m, k), device="cuda", dtype=torch.float16)
c = torch.empty((m, n), device="cuda", dtype=torch.float16)
for _ in range(300): ext.gemm_half_q_half(a, h, c)
torch.cuda.synchronize()
groups = (n + 127) // 128
for ks in [int(v) for v in sys.argv[4:]]:
    tr = torch.zeros((groups * ks, 8), dtype=torch.int64, device="cuda")
    os.environ["EXL2_SKINNY_SPLITK"] = str(ks)
    for _ in range(20): ext.gemm_half_q_half(a, h, c)
    os.environ["EXL2_SKINNY_TRACE_PTR"] = str(tr.data_ptr())
    ext.gemm_half_q_half(a, h, c); torch.cuda.synchronize()
    os.environ.pop("EXL2_SKINNY_TRACE_PTR")
    t = tr.cpu().numpy().astype(np.float64) * 0.01            # us
    t0 = t[:, 0].min()
    rel = t - t0
    names = ["entry", "map staged", "K loop done", "partial stored", "ticket back", "reduce done", "end"]
    print(f"== K {k} N {n} M {m} ks {ks}: {groups * ks} workgroups; launch span {rel[:, :7].max():.2f} us")
    for i, nm in enumerate(names):
        v = rel[:, i][t[:, i] > 0]
        if len(v): print(f"  {nm:15s} n {len(v):4d}  min {v.min():6.2f}  p50 {np.median(v):6.2f}  p90 {np.percentile(v, 90):6.2f}  max {v.max():6.2f}")
    own = t[:, 2] - t[:, 1]
    print(f"  K loop (own)    p50 {np.median(own):6.2f}  max {own.max():6.2f};  entry->map p50 {np.median(t[:, 1] - t[:, 0]):5.2f}; store p50 {np.median(t[:, 3] - t[:, 2]):5.2f}; ticket p50 {np.median(t[:, 4] - t[:, 3]):5.2f}")
    last = t[:, 5] > 0
    if last.any(): print(f"  last arrivers: reduce p50 {np.median((t[:, 5] - t[:, 4])[last]):5.2f}  epilogue p50 {np.median((t[:, 6] - t[:, 5])[last]):5.2f}")
