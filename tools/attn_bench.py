#!/usr/bin/env python3
"""Decode attention kernel timing per shape (fused one-launch kernel), HIP events around 200 back-to-back launches."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from exllamav2_amd.ext import ext_c as ext

def run(nh, kvh, hd, ctx, pages=80, quiet=False):
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
    n = 60 if quiet else 200
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    kvb = 2 * ctx * kvh * hd * 2
    if quiet:
        return us
    print(json.dumps({"heads": nh, "kv_heads": kvh, "head_dim": hd, "ctx": ctx, "us": round(us, 2), "kv_MB": round(kvb / 1e6, 2),
                      "GBs": round(kvb / us / 1e3, 1)}), flush=True)
    return us

if __name__ == "__main__":
    if "--sweep" in sys.argv:
        # splits x keys-per-split sweep at long contexts (EXL2_ATT_NSPLIT_MAX / EXL2_ATT_KPS are read per launch)
        for nh, kvh in ((32, 32), (64, 8)):
            for ctx in (2000, 8000, 16000):
                for nsplit, kps in ((0, 0), (16, 64), (32, 64), (32, 128), (64, 64), (64, 128), (64, 256)):
                    os.environ.pop("EXL2_ATT_NSPLIT_MAX", None); os.environ.pop("EXL2_ATT_KPS", None)
                    if nsplit: os.environ["EXL2_ATT_NSPLIT_MAX"] = str(nsplit); os.environ["EXL2_ATT_KPS"] = str(kps)
                    print(json.dumps({"nsplit_max": nsplit or "policy", "kps": kps or "default"}), end=" ", flush=True)
                    run(nh, kvh, 128, ctx)
        sys.exit(0)
    if "--sweep2" in sys.argv:
        # finer grid behind the split policy of csrc/attn.hip: best (splits, keys per split) per shape and context
        for nh, kvh, pages in ((32, 32, 140), (64, 8, 140), (32, 8, 140)):
            for ctx in (500, 1000, 2000, 4000, 8000, 16000, 32000):
                best = None
                for nsplit in (4, 8, 16, 24, 32, 48, 64):
                    for kps in (64, 128, 256, 512, 1024):
                        if nsplit * kps < ctx / 4 and nsplit < 64: continue          # (far too few splits for this context)
                        os.environ["EXL2_ATT_NSPLIT_MAX"] = str(nsplit); os.environ["EXL2_ATT_KPS"] = str(kps)
                        us = run(nh, kvh, 128, ctx, pages=pages, quiet=True)
                        if best is None or us < best[0]: best = (us, nsplit, kps)
                        print(json.dumps({"heads": nh, "kv_heads": kvh, "ctx": ctx, "nsplit_max": nsplit, "kps": kps, "us": round(us, 2)}), flush=True)
                os.environ.pop("EXL2_ATT_NSPLIT_MAX", None); os.environ.pop("EXL2_ATT_KPS", None)
                pol = run(nh, kvh, 128, ctx, pages=pages, quiet=True)
                print(json.dumps({"heads": nh, "kv_heads": kvh, "ctx": ctx, "BEST": {"us": round(best[0], 2), "nsplit_max": best[1], "kps": best[2]}, "policy_us": round(pol, 2)}), flush=True)
        sys.exit(0)
    for nh, kvh in ((32, 32), (64, 8), (32, 8)):
        for ctx in (64, 2000, 16000):
            run(nh, kvh, 128, ctx)
