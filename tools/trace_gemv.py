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
set_trace.argtypes = [ctypes.c_void_p, ctypes.c_int]; set_trace.restype = None
NAMES = ["entry", "prologue DMA issued", "DMA landed + barrier", "shuffle done", "barrier1 passed", "main run done",
         "minor runs done", "barrier2 passed", "end", "  (ring fill issued)", "  (scale table done)", "  (rms done)", "  (arguments loaded)"]
ORDER = [0, 12, 1, 2, 9, 10, 11, 3, 4, 5, 6, 7, 8]
TICK_US = 0.01


def run(k, n, recipe, reps=2, m=1, norm=False):
    gen = torch.Generator(device="cuda"); gen.manual_seed(0)
    hs, keep = [], []
    for _ in range(reps + 1):
        w = synth_linear(k, n, recipe, "cuda", gen); keep.append(w)
        hs.append(ext.make_q_matrix_from_dict(w, none_tensor))
    a = torch.randn((m, k), device="cuda", dtype=torch.float16)
    c = torch.empty((m, n), device="cuda", dtype=torch.float16)
    ext.gemm_half_q_half(a, hs[0], c); torch.cuda.synchronize()
    buf = torch.zeros((4096 * 16 * 16,), dtype=torch.int64, device="cuda")
    for h in hs[1:]:
        buf.zero_(); torch.cuda.synchronize()
        set_trace(buf.data_ptr(), 0)
        ext.gemm_half_q_half(a, h, c); torch.cuda.synchronize()
        report(buf, f"shape {k}x{n} m={m}")
    set_trace(None, 0)
    for h in hs: ext.free_q_matrix(h)


def report(buf, title):
    if True:
        t = buf.cpu().numpy().reshape(4096, 16, 16)
        used = t[:, :, 0] != 0
        tt = t[used][:, :13].astype(np.float64) * TICK_US    # [waves, 13] in us
        t0 = tt[:, 0].min()
        print(f"{title}: waves {tt.shape[0]}  (us since the first wave's entry: median / p10 / p90 / max)")
        for i in ORDER:
            col = tt[:, i] - t0
            print(f"  {NAMES[i]:20s} {np.median(col):7.2f} {np.percentile(col,10):7.2f} {np.percentile(col,90):7.2f} {col.max():7.2f}")
        hw = t[used][:, 13]
        xcc = (hw >> 32) & 0xF
        cu = (hw >> 8) & 0xF; sh = (hw >> 12) & 0x1; se = (hw >> 13) & 0x7
        cuid = ((xcc * 8 + se) * 2 + sh) * 16 + cu
        ids, counts = np.unique(cuid, return_counts=True)
        hist = np.bincount(counts)
        print(f"  placement: {len(ids)} distinct CUs; waves per CU histogram " + ", ".join(f"{n}w:{c}" for n, c in enumerate(hist) if c))
        endc = np.array([tt[cuid == i, 8].max() - t0 for i in ids])
        for nwaves in np.unique(counts):
            sel = endc[counts == nwaves]
            print(f"    CUs with {nwaves:2d} waves: last end median {np.median(sel):6.2f} max {sel.max():6.2f}")
        life = tt[:, 8] - tt[:, 0]
        print(f"  wave lifetime        median {np.median(life):6.2f}  p90 {np.percentile(life,90):6.2f}  max {life.max():6.2f};"
              f" kernel span {tt[:, 8].max() - t0:6.2f} us")


def run_mlp(hidden=4096, inter=11008, recipe=([4], [1.0], 128)):
    """gate+up (RMSNorm fused) and down (SiLU*up fused) launches of q_mlp_forward_"""
    gen = torch.Generator(device="cuda"); gen.manual_seed(1)
    ws = [synth_linear(hidden, inter, recipe, "cuda", gen), synth_linear(hidden, inter, recipe, "cuda", gen),
          synth_linear(inter, hidden, recipe, "cuda", gen)]
    hs = [ext.make_q_matrix_from_dict(w, none_tensor) for w in ws]
    norm = torch.ones((hidden,), device="cuda", dtype=torch.float16)
    ts = torch.empty((16, hidden), device="cuda", dtype=torch.float16)
    ta = torch.empty((16, inter), device="cuda", dtype=torch.float16)
    tb = torch.empty((16, inter), device="cuda", dtype=torch.float16)
    mlp = ext.make_q_mlp(norm, none_tensor, True, 1e-5, hs[0], hs[1], hs[2], ts, ta, tb, none_tensor, 16, False, True,
                         none_tensor, none_tensor, False, False)
    x = torch.randn((1, 1, hidden), device="cuda", dtype=torch.float16)
    buf = torch.zeros((4096 * 16 * 16,), dtype=torch.int64, device="cuda")
    ext.q_mlp_forward_(mlp, x); torch.cuda.synchronize()
    for which, name in ((0, "gate+up (rmsnorm prologue)"), (1, "down (silu*up prologue)")):
        buf.zero_(); torch.cuda.synchronize()
        set_trace(buf.data_ptr(), which)
        ext.q_mlp_forward_(mlp, x); torch.cuda.synchronize()
        report(buf, f"mlp {hidden}->{inter}: {name}")
    set_trace(None, 0)


if __name__ == "__main__":
    run_mlp()
    sys.exit(0)
    r4 = ([4], [1.0], 128)
    run(4096, 4096, r4)
    run(4096, 11008, r4)
    run(11008, 4096, r4)
    run(4096, 32000, r4, 1)
