#!/usr/bin/env python3
"""MFMA flash-prefill attention (csrc/attn_prefill.hip) at BASELINE configs[2]'s shape (8 x 2048 tokens, 32 heads, hd 128,
FP16 contiguous cache), one layer: time per call, causal TFLOP/s (2 * 2 * b * H * s^2 / 2 * hd), next to torch SDPA on the
same tensors (what `_attn_torch`, attn.py:869-937, costs on this GPU; a yardstick, not on the product path)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from exllamav2_amd.ext import ext_c as ext


def timed(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    dev = "cuda:0"
    for (b, s, H, KVH, hd) in ((8, 2048, 32, 32, 128), (1, 2048, 32, 32, 128), (8, 2048, 64, 8, 128), (4, 512, 32, 4, 64)):
        q = torch.randn((b, s, H, hd), dtype=torch.float16, device=dev)
        k = torch.randn((b, s, KVH, hd), dtype=torch.float16, device=dev)
        v = torch.randn((b, s, KVH, hd), dtype=torch.float16, device=dev)
        out = torch.empty_like(q)
        ms = timed(lambda: ext.flash_prefill(q, k, v, out, None, None, len_const=0, len_offset=s))
        flops = 2 * 2 * b * H * s * s / 2 * hd
        g = H // KVH
        qh, kh, vh = q.transpose(1, 2), k.transpose(1, 2).repeat_interleave(g, 1), v.transpose(1, 2).repeat_interleave(g, 1)
        ms_t = timed(lambda: torch.nn.functional.scaled_dot_product_attention(qh, kh, vh, is_causal=True))
        ref = torch.nn.functional.scaled_dot_product_attention(qh, kh, vh, is_causal=True).transpose(1, 2)
        err = (out.float() - ref.float()).abs().max().item()
        print(json.dumps({"shape": f"b{b} s{s} H{H} KVH{KVH} hd{hd}", "flash_prefill_ms": round(ms, 3),
                          "tflops_causal": round(flops / ms / 1e9, 1), "torch_sdpa_ms": round(ms_t, 3),
                          "max_abs_diff_vs_sdpa": round(err, 4)}))


if __name__ == "__main__":
    main()
