#!/usr/bin/env python3
"""In-kernel timeline of a producer / consumer pair of the chained decode (EXL2_TRACE build), serial chain (one graph) against
the overlapped chain (EXL2_CHAIN_OVERLAP=1: two graphs side by side), 2-layer Llama-2-7B-width model, graph replay.
Stamps: 100 MHz s_memrealtime shared by all XCDs; times are relative to the PRODUCER's first wave entry."""
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
set_trace = lib.dll.exl2_debug_set_flat_trace
set_trace.argtypes = [ctypes.c_void_p, ctypes.c_int]; set_trace.restype = None
NAMES = {0: "entry", 12: "split computed", 1: "tables issued", 9: "ring issued", 10: "at the wait", 11: "past the wait",
         3: "prologue done", 4: "barrier 1 passed", 5: "stream done", 7: "barrier 2 passed", 8: "end"}
ORDER = [0, 12, 1, 9, 10, 11, 3, 4, 5, 7, 8]
PAIRS = {"o -> gate|up": 5, "gate|up -> down": 6, "down -> head": 7, "q|k|v (layer 1)": 4}


def table(t, t0, title):
    used = t[:, :, 0] != 0
    tt = t[used].astype(np.float64) * 0.01
    print(f"  {title}: {tt.shape[0]} waves   (median / p10 / p90 / max, us after the producer's first entry)")
    for i in ORDER:
        col = tt[:, i]; col = col[col > 0] - t0
        if col.size: print(f"    {NAMES[i]:18s} {np.median(col):7.2f} {np.percentile(col,10):7.2f} {np.percentile(col,90):7.2f} {col.max():7.2f}")
    return tt


def main():
    cfg = ExLlamaV2Config.llama2_7b(max_seq_len=512, max_input_len=32)
    cfg.num_hidden_layers = 2
    ck = synth_checkpoint(cfg, "cuda:0", recipe="4.0bpw", seed=0)
    model = ExLlamaV2(cfg, device="cuda:0", ext=ext).load(ck)
    cache = ExLlamaV2Cache(model, batch_size=1)
    buf = torch.zeros((2 * 256 * 16 * 16,), dtype=torch.int64, device="cuda")
    for name, first in PAIRS.items():
        for mode in ("0", "1"):
            os.environ["EXL2_CHAIN_OVERLAP"] = mode
            dec = GreedyGraphDecoder(model, cache, batch_size=1)
            assert dec.chain is not None and ("flags" in dec.chain) == (mode == "1")
            # capture() = one eager warm-up step (9 chained q_gemm launches) + the captured step: stamp launches of the latter
            set_trace(buf.data_ptr(), 9 + first)
            dec.capture()
            dec.reset(torch.tensor([1]), 0)
            dec.run(3); torch.cuda.synchronize()
            buf.zero_(); torch.cuda.synchronize()
            dec.run(1); torch.cuda.synchronize()
            t = buf.cpu().numpy().reshape(2, -1, 16, 16)
            p = t[0][t[0][:, :, 0] != 0]
            t0 = p[:, 0].min() * 0.01
            print(f"== {name}: {'overlapped (two graphs)' if mode == '1' else 'serial chain (one graph)'}")
            pt = table(t[0], t0, "producer")
            ct = table(t[1], t0, "consumer")
            print(f"    producer last end {pt[:, 8].max() - t0:6.2f}   consumer first decode-ready (barrier 1) {ct[:, 4][ct[:, 4] > 0].min() - t0:6.2f}   consumer last end {ct[:, 8].max() - t0:6.2f}")
            set_trace(None, 0)
            dec.free()
    model.unload()


if __name__ == "__main__":
    main()
