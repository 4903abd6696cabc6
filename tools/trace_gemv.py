#!/usr/bin/env python3
"""In-kernel timeline of the streaming q_gemm (EXL2_TRACE build): where does a launch spend its microseconds?

Build: hipcc ... -DEXL2_TRACE -o exllamav2_amd/libexl2_hip_trace.so (tools/build_trace.sh).  Timestamps are the 100 MHz
s_memrealtime counter (10 ns ticks, shared by all XCDs), one set per wave."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from exllamav2_amd import _lib
from exllamav2_amd.ext import ExtC, none_tensor
from exllamav2_amd.synth import synth_linear

lib = _lib.Lib(os.path.join(ROOT, "exllamav2_amd", "libexl2_hip_trace.so"))
ext = ExtC(lib)
set_trace = lib.dll.exl2_debug_set_trace
set_trace.argtypes = [ctypes.c_void_p]; set_trace.restype = None
NAMES = ["entry", "ring fill issued", "tables issued", "staging done", "barrier1 passed", "main run done",
         "minor runs done", "barrier2 passed", "end"]
TICK_US = 0.01


def run(k, n, recipe, reps=2, m=1):
    gen = torch.Generator(device="cuda"); gen.manual_seed(0)
    hs, keep = [], []
    for _ in range(reps + 1):
        w = synth_linear(k, n, recipe, "cuda", gen); keep.append(w)
        hs.append(ext.make_q_matrix_from_dict(w, none_tensor))
    a = torch.randn((m, k), device="cuda", dtype=torch.float16)
    c = torch.empty((m, n), device="cuda", dtype=torch.float16)
    ext.gemm_half_q_half(a, hs[0], c); torch.cuda.synchronize()
    buf = torch.zeros((4096 * 16 * 16,), dtype=torch.int64, device="cuda")
    set_trace(buf.data_ptr())
    for h in hs[1:]:
        buf.zero_(); torch.cuda.synchronize()
        ext.gemm_half_q_half(a, h, c); torch.cuda.synchronize()
        t = buf.cpu().numpy().reshape(4096, 16, 16)
        used = t[:, :, 0] != 0
        tt = t[used][:, :9].astype(np.float64) * TICK_US     # [waves, 9] in us
        t0 = tt[:, 0].min()
        print(f"shape {k}x{n} m={m}: waves {tt.shape[0]}  (us since the first wave's entry: median / p10 / p90 / max)")
        for i in range(9):
            col = tt[:, i] - t0
            print(f"  {NAMES[i]:20s} {np.median(col):7.2f} {np.percentile(col,10):7.2f} {np.percentile(col,90):7.2f} {col.max():7.2f}")
        life = tt[:, 8] - tt[:, 0]
        print(f"  wave lifetime        median {np.median(life):6.2f}  p90 {np.percentile(life,90):6.2f}  max {life.max():6.2f};"
              f" kernel span {tt[:, 8].max() - t0:6.2f} us")
    set_trace(None)
    for h in hs: ext.free_q_matrix(h)


if __name__ == "__main__":
    r4 = ([4], [1.0], 128)
    run(4096, 4096, r4)
    run(4096, 11008, r4)
    run(11008, 4096, r4)
    run(4096, 32000, r4, 1)
