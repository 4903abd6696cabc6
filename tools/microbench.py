#!/usr/bin/env python3
"""Kernel-level microbenchmarks on one GPU: achieved HBM GB/s of the q_gemm GEMV per shape / bit mix / M, with the
reference's method (tests/test_gemv.py:84-128): rotate enough distinct matrices that the working set exceeds the
256 MB Infinity Cache, so the number is an HBM number.  Also a device-to-device copy for the achievable ceiling."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from exllamav2_amd.ext import ext_c, none_tensor  # noqa: E402
from exllamav2_amd.synth import synth_linear  # noqa: E402


def bench_copy(nbytes=1 << 30, iters=10):
    a = torch.empty(nbytes // 4, dtype=torch.float32, device="cuda").normal_()
    b = torch.empty_like(a)
    b.copy_(a); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): b.copy_(a)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    return 2 * nbytes / (ms * 1e-3) / 1e9


def bench_gemv(k, n, recipe, m=1, min_bytes=600e6, iters=5, act_order=True):
    gen = torch.Generator(device="cuda"); gen.manual_seed(0)
    mats, handles = [], []
    total = 0
    while total < min_bytes:
        w = synth_linear(k, n, recipe, "cuda", gen, act_order=act_order)
        h = ext_c.make_q_matrix_from_dict(w, none_tensor)
        mats.append(w); handles.append(h)
        total += ext_c.q_matrix_info(h)["bytes"]
    a = torch.randn((m, k), device="cuda", dtype=torch.float16)
    c = torch.empty((m, n), device="cuda", dtype=torch.float16)
    for h in handles: ext_c.gemm_half_q_half(a, h, c)
    torch.cuda.synchronize()
    # replay the rotation from a HIP graph so the number is GPU time (kernel + launch boundary), not python overhead
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        ext_c.graph_begin_capture(st.cuda_stream)
        for h in handles: ext_c.gemm_half_q_half(a, h, c)
        g = ext_c.graph_end_capture(st.cuda_stream)
        ext_c.graph_launch(g, st.cuda_stream)
        st.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(iters): ext_c.graph_launch(g, st.cuda_stream)
        e1.record(st)
        st.synchronize()
    ext_c.graph_free(g)
    ms = e0.elapsed_time(e1) / (iters * len(handles))
    per = total / len(handles)
    for h in handles: ext_c.free_q_matrix(h)
    return {"k": k, "n": n, "m": m, "recipe": str(recipe), "mats": len(handles), "us": round(ms * 1e3, 2),
            "GBs": round(per / (ms * 1e-3) / 1e9, 1), "frac_8TBs": round(per / (ms * 1e-3) / 8e12, 3)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    args = ap.parse_args()
    out = {"copy_GBs": round(bench_copy(), 1)}
    print(json.dumps(out), flush=True)
    r4 = ([4], [1.0], 128)
    shapes = [(4096, 4096), (4096, 11008), (11008, 4096), (4096, 32000)]
    for k, n in shapes:
        print(json.dumps(bench_gemv(k, n, r4)), flush=True)
    if not args.quick:
        for rec in (([5, 4], [0.1, 0.9], 128), ([4], [1.0], 32), ([6], [1.0], 128), ([8, 4], [0.05, 0.95], [32, 128]),
                    ([3, 2], [0.1, 0.9], 64)):
            print(json.dumps(bench_gemv(4096, 11008, rec)), flush=True)
        for m in (2, 4, 8, 16):
            print(json.dumps(bench_gemv(4096, 11008, r4, m=m)), flush=True)
        print(json.dumps(bench_gemv(4096, 11008, r4, act_order=False)), flush=True)


if __name__ == "__main__":
    main()
