#!/usr/bin/env python3
"""Decode attention kernel timing per shape (fused one-launch kernel), HIP events around 200 back-to-back launches."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from exllamav2_amd.ext import ext_c as ext

def run(nh, kvh, hd, ctx, pages=80):
    dev = "cuda"
    q = torch.randn((1, 1, nh, hd), device=dev, dtype=torch.float16)
    kn = torch.randn((1, 1, kvh, hd), device=dev, dtype=torch.float16)
    vn = torch.randn((1, 1, kvh, hd), device=dev, dtype=torch.float16)
    kc = torch.randn((pages, 256, kvh, hd), device=dev, dtype=torch.float16)
    vc = torch.randn((pages, 256, kvh, hd), device=dev, dtype=torch.float16)
    out = torch.empty_like(q)
    sin = torch.zeros((pages * 256, hd), device=dev, dtype=torch.float16); cos = torch.ones_like(sin)
    sl = torch.tensor([ctx], dtype=torch.int32, device=dev)
    bt = torch.arange(pages, dtype=torch.int32, device=dev).view(1, pages)
    scratch = torch.zeros((ext.paged_attn_scratch_bytes(nh, hd, 64) // 4 + 16,), dtype=torch.float32, device=dev)
    cnt = torch.zeros((4096,), dtype=torch.int32, device=dev)
    f = lambda: ext.attn_decode_fused(q, kn, vn, kc, vc, out, sin, cos, sl, bt, 0, 2, scratch, cnt)
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 200
    kvb = 2 * ctx * kvh * hd * 2
    print(json.dumps({"heads": nh, "kv_heads": kvh, "head_dim": hd, "ctx": ctx, "us": round(us, 2), "kv_MB": round(kvb / 1e6, 2),
                      "GBs": round(kvb / us / 1e3, 1)}), flush=True)

if __name__ == "__main__":
    for nh, kvh in ((32, 32), (64, 8), (32, 8)):
        for ctx in (64, 2000, 16000):
            run(nh, kvh, 128, ctx)
