#!/usr/bin/env python3
"""In-kernel timeline of the lean chained decode launches (csrc/qgemv_lean.hip; EXL2_TRACE build, tools/build_trace.sh): one
decode step of a 2-layer Llama-2-7B-width model; the q|k|v, o, gate|up, down launches of layer 1 and the head launch are
stamped.  Timestamps: 100 MHz s_memrealtime (10 ns ticks, shared by all XCDs), one set per wave."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from exllamav2_amd import _lib
from exllamav2_amd.ext import ExtC
from exllamav2_amd.config import ExLlamaV2Config
from exllamav2_amd.model import ExLlamaV2, GreedyGraphDecoder
from exllamav2_amd.cache import ExLlamaV2Cache
from exllamav2_amd.synth import synth_checkpoint

lib = _lib.Lib(os.path.join(ROOT, "exllamav2_amd", "libexl2_hip_trace.so"))
ext = ExtC(lib)
set_trace = lib.dll.exl2_debug_set_lean_trace
set_trace.argtypes = [ctypes.c_void_p, ctypes.c_int]; set_trace.restype = None
NAMES = ["arguments arrived", "arguments unpacked", "weights requested", "rest unpacked, staged copies issued", "first part + staged copies landed (fence load)",
         "decoded, partials written", "barrier passed", "end (finalising waves)"]
TICK_US = 0.01
MATS, WGS, WAVES = 4, 2048, 16


def report(buf, title):
    t = buf.cpu().numpy().reshape(MATS * WGS, WAVES, 8)
    used = t[:, :, 0] != 0
    tt = t[used].astype(np.float64) * TICK_US
    if tt.shape[0] == 0:
        print(f"{title}: no waves stamped"); return
    t0 = tt[:, 0].min()
    print(f"{title}: waves {tt.shape[0]}  (us since the first wave's entry: median / p10 / p90 / max)")
    for i in range(8):
        col = tt[:, i]
        col = col[col > 0] - t0
        if col.size: print(f"  {NAMES[i]:42s} {np.median(col):7.2f} {np.percentile(col,10):7.2f} {np.percentile(col,90):7.2f} {col.max():7.2f}   n={col.size}")
    last = np.where(tt[:, 7] > 0, tt[:, 7], tt[:, 6])
    print(f"  kernel span {last.max() - t0:6.2f} us")
    # per wave: how long between its own entry and its own milestones (median)
    own = tt - tt[:, :1]
    print("  per wave, since ITS first stamp (median): " + ", ".join(f"[{i}] {np.median(own[:, i][tt[:, i] > 0]):.2f}" for i in range(1, 7)))
    # entry times of workgroups: dispatch rounds
    we = t[:, :, 0].max(axis=1).astype(np.float64) * TICK_US
    we = we[we > 0] - t0
    print(f"  workgroup entry: median {np.median(we):.2f}, p90 {np.percentile(we, 90):.2f}, max {we.max():.2f} us ({we.size} workgroups)")


def main():
    batch = int(os.environ.get("B", "1"))
    cfg = ExLlamaV2Config.llama2_7b(max_seq_len=512, max_input_len=32)
    cfg.num_hidden_layers = 2
    ck = synth_checkpoint(cfg, "cuda:0", recipe=os.environ.get("RECIPE", "4.0bpw"), seed=0)
    model = ExLlamaV2(cfg, device="cuda:0", ext=ext).load(ck)
    cache = ExLlamaV2Cache(model, batch_size=batch)
    dec = GreedyGraphDecoder(model, cache, batch_size=batch)
    assert dec.chain is not None
    dec.reset(torch.tensor([1] * batch), 0)
    dec.run(3, use_graph=False); torch.cuda.synchronize()
    buf = torch.zeros((MATS * WGS * WAVES * 8,), dtype=torch.int64, device="cuda")
    for which, name in ((4, "q|k|v (A_NORM_PRE)"), (5, "o (A_DIRECT, residual, chain-out)"), (6, "gate|up (A_NORM_PRE, pair)"),
                        (7, "down (A_DIRECT, residual, chain-out)"), (8, "head (A_NORM_PRE)")):
        for rep in range(2):
            buf.zero_(); torch.cuda.synchronize()
            set_trace(buf.data_ptr(), which)
            dec.run(1, use_graph=False); torch.cuda.synchronize()
        report(buf, name)
    set_trace(None, 0)


if __name__ == "__main__":
    main()
